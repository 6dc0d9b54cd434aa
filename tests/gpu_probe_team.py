"""Ad-hoc GPU probe (not a test): where the lane-team engine (16 lanes per message / pair) hands over to the one-lane-per-item kernels.
Run against a library built with -DBLS_EXPERIMENTS (the MI355_BLS_TEAM_*_MAX knobs exist only there):
    BLS_EXTRA_FLAGS=-DBLS_EXPERIMENTS BLS_OUT=variants/exp.so nim-blscurve_amd/build.sh
    MI355_BLS_LIB=nim-blscurve_amd/variants/exp.so python tests/gpu_probe_team.py
Prints, per batch size, the cofactor-clearing and Miller-line stage times of a latency-mode context with the engine forced on and off."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import torch
    import __graft_entry__ as ge
    import bench
    m = ge.load_package()
    dev = torch.device("cuda", 0)
    rnd = hashlib.sha256(b"Mr F was here").digest()
    gen = m.BatchedBLSVerifierCache.init(max_sets=32768)
    base = bench.sign_records(m, gen, dev, range(32768))
    for nn in (1024, 2048, 4096, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 28672):
        c = m.BatchedBLSVerifierCache.init(max_sets=nn)
        best = None
        for _ in range(3):
            assert c.verify_device(base.data_ptr(), nn, rnd)
            t, k = c.timings(), c.kernel_timings()
            row = (k["k_hash_clear"], t["miller_lines"], t["total"])
            best = row if best is None or row[2] < best[2] else best
        print("%s n=%d clear %.3f lines %.3f total %.3f" % (sys.argv[1], nn, *best), flush=True)
        c.close()
else:
    for tag, v in (("team", "1000000"), ("lane", "0")):
        env = dict(os.environ, MI355_BLS_TEAM_CLEAR_MAX=v, MI355_BLS_TEAM_LINES_MAX=v)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag], env=env)
