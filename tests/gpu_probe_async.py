"""Ad-hoc GPU probe: host time of submit / wait (not a test)."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
n = 65536
streams = [torch.cuda.Stream() for _ in range(4)]
gen = m.BatchedBLSVerifierCache.init(max_sets=n)
d_sets = bench.sign_records(m, gen, dev, range(n))
caches = [m.BatchedBLSVerifierCache.init(max_sets=n) for _ in range(4)]
rnd = hashlib.sha256(b"Mr F was here").digest()
for c, s in zip(caches, streams):
    assert c.verify_device(d_sets.data_ptr(), n, rnd, s.cuda_stream)
torch.cuda.synchronize()
for rep in range(2):
    ts = []
    t0 = time.perf_counter()
    for i in range(4):
        a = time.perf_counter(); caches[i].submit_device(d_sets.data_ptr(), n, rnd, streams[i].cuda_stream, after=caches[(i - 1) % 4]); ts.append(time.perf_counter() - a)
    tw = []
    for i in range(4):
        a = time.perf_counter(); assert caches[i].wait(); tw.append(time.perf_counter() - a)
    print("submit ms", [round(x * 1e3, 2) for x in ts], "wait ms", [round(x * 1e3, 2) for x in tw], "total", round((time.perf_counter() - t0) * 1e3, 2))
