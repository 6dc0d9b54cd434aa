# On the GPU box: launch-by-launch k_tail stamps of the debug variant nim-blscurve_amd/variants/rowdbg.so (BLS_OUT=variants/rowdbg.so BLS_EXTRA_FLAGS="-DBLS_TAIL_CLOCK -DBLS_C12_ROW" bash build.sh):
# duration (s_memrealtime), SIMD / CU / SE of every wave (HW_ID), LDS allocation base - the data behind profiles/r04_ab/row_engine.txt
R=$GRAFT_REPO_ROOT
cp $R/nim-blscurve_amd/libblscurve_mi355x.so /tmp/keep.so
cp $R/nim-blscurve_amd/variants/rowdbg.so $R/nim-blscurve_amd/libblscurve_mi355x.so; touch $R/nim-blscurve_amd/libblscurve_mi355x.so
python3 $R/tests/gpu_probe_aux.py fav b4096 2>&1 | grep -A1 "k_tail mode . wave 0" | grep -v "^--" | tail -36
cp /tmp/keep.so $R/nim-blscurve_amd/libblscurve_mi355x.so
