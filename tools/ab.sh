# A/B of two prebuilt libraries on the same box: nim-blscurve_amd/variants/{a,b}.so
R=$GRAFT_REPO_ROOT
for v in c al c al; do
  cp $R/nim-blscurve_amd/variants/$v.so $R/nim-blscurve_amd/libblscurve_mi355x.so
  touch $R/nim-blscurve_amd/libblscurve_mi355x.so
  echo "== $v"; timeout 300 bash $R/tools/kstats.sh 2>&1 | grep -E "k_lines|k_lineprod\(|k_hash|k_pkmul|k_sig_bucket" | awk '{print $(NF-1), $NF, $3}' | tr '\n' ';'; echo
  timeout 300 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-aux 2>&1 | tail -1 | cut -c70-130
done
