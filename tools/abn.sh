# Same-box A/B/N of prebuilt libraries nim-blscurve_amd/variants/<name>.so (box-to-box variance is a few percent, so
# optimisations are judged on one box, interleaved).  usage (on the GPU box): [WHAT=bench|msm|lat] bash tools/abn.sh ROUNDS name1 name2 ...
R=$GRAFT_REPO_ROOT; rounds=$1; shift
# a variant is selected with MI355_BLS_LIB (nim-blscurve_amd/__init__.py): the shipped library is never overwritten
for r in $(seq $rounds); do for v in "$@"; do
  export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/$v.so
  echo -n "$v "
  case "${WHAT:-bench}" in
    msm) timeout 300 python3 $R/tests/gpu_probe_aux.py msm 2>/dev/null | tail -2 | tr '\n' ' '; echo ;;
    lat) timeout 300 python3 $R/tests/gpu_probe_lat.py 2>/dev/null | grep -E "FAV n=32768|batch n=(64|4096|65536):" | cut -c1-60 | tr '\n' ';'; echo ;;
    *) timeout 300 python3 $R/bench.py --steps 30 --warmup 4 --no-cpu --no-aux 2>/dev/null | head -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['ms_one_caller'],2), {k:round(v,3) for k,v in d['kernel_ms_alone'].items()}, {k:round(v,3) for k,v in d.get('tail_ms_alone',{}).items()})" ;;
  esac
done; done
unset MI355_BLS_LIB
