"""Ad-hoc GPU probe: host-buffer entry point (PCIe-inclusive) against the device-resident one (not a test)."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
n = 65536
gen = m.BatchedBLSVerifierCache.init(max_sets=n)
d_sets = bench.sign_records(m, gen, dev, range(n))
host = bytes(d_sets.cpu().numpy())
cache = m.BatchedBLSVerifierCache.init(max_sets=n)
rnd = hashlib.sha256(b"Mr F was here").digest()
for rep in range(3):
    t0 = time.perf_counter(); ok = m.batchVerify(cache, host, rnd); t1 = time.perf_counter()
    kern = cache.timings()["total"]
    t2 = time.perf_counter(); ok2 = cache.verify_device(d_sets.data_ptr(), n, rnd); t3 = time.perf_counter()
    print("host-buffer call %.2f ms (kernels %.2f ms)   device-resident call %.2f ms" % ((t1 - t0) * 1e3, kern, (t3 - t2) * 1e3), ok, ok2)
