# same-box A/B of the pipelined bench: nim-blscurve_amd/variants/{c,al}.so, 3 runs each, interleaved
R=$GRAFT_REPO_ROOT
for v in c al c al c al; do
  cp $R/nim-blscurve_amd/variants/$v.so $R/nim-blscurve_amd/libblscurve_mi355x.so; touch $R/nim-blscurve_amd/libblscurve_mi355x.so
  echo -n "$v "; timeout 300 python3 $R/bench.py --steps 30 --warmup 4 --no-cpu --no-aux 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['value']))"
done
