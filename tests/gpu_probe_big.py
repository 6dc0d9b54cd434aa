"""Ad-hoc GPU probe: one blocking call at larger batch sizes (not a test)."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
base = None
for n in (65536, 131072, 262144, 1048576):
    gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
    if base is None:
        base = bench.sign_records(m, gen, dev, range(65536))
    gen.close()
    d = base.repeat(n // 65536)
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    assert cache.verify_device(d.data_ptr(), n, rnd)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); ok = cache.verify_device(d.data_ptr(), n, rnd); ts.append((time.perf_counter() - t0) * 1e3)
        assert ok
    print("n=%d: blocking call %.2f ms, %.2f M verifications/s" % (n, min(ts), n / min(ts) / 1e3), flush=True)
    cache.close()
    del d
