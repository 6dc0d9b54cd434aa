# On the GPU box: launch-by-launch k_tail stamps of the debug variant nim-blscurve_amd/variants/rowdbg.so (BLS_OUT=variants/rowdbg.so BLS_EXTRA_FLAGS="-DBLS_TAIL_CLOCK -DBLS_C12_ROW" bash build.sh):
# duration (s_memrealtime), SIMD / CU / SE of every wave (HW_ID), LDS allocation base - the data behind profiles/r04_ab/row_engine.txt
R=$GRAFT_REPO_ROOT
# a variant is selected with MI355_BLS_LIB (nim-blscurve_amd/__init__.py): the shipped library is never overwritten
export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/rowdbg.so
python3 $R/tests/gpu_probe_aux.py fav b4096 2>&1 | grep -A1 "k_tail mode . wave 0" | grep -v "^--" | tail -36
unset MI355_BLS_LIB
