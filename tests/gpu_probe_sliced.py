"""One blocking batchVerify of 2^20 tuples through a context of 65 536 sets (16 slices, pipelined over the context's three
workspaces): ms per call against 16 x the pipelined per-batch time of bench.py; the GT value equals the unsliced call's
(tests/test_gpu_capacity.py covers the small sizes).  usage: python3 tests/gpu_probe_sliced.py [log2 n]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
dev = torch.device("cuda", 0)
gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
parts = [bench.sign_records(m, gen, dev, range(3_000_000 + i * 65536, 3_000_000 + (i + 1) * 65536)) for i in range(n // 65536)]
d = torch.cat(parts)
gen.close()
del parts
rnd = hashlib.sha256(b"sliced").digest()
small = m.BatchedBLSVerifierCache.init(max_sets=65536, numThreads=4096)
small.set_cooperative(False)
assert small.verify_device(d.data_ptr(), n, rnd) is True          # warm-up (creates the lanes)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for i in range(reps):
    assert small.verify_device(d.data_ptr(), n, rnd) is True
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
gt = small.fetch(4, 576)
print("sliced: n = 2^%d through a 65536-set context: %.1f ms per blocking call = %.2f ms per 65536 tuples, %.2f M verifications/s" % (lg, dt * 1e3, dt * 1e3 * 65536 / n, n / dt / 1e6))
if lg <= 18:
    big = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=4096)
    assert big.verify_device(d.data_ptr(), n, rnd) is True and big.fetch(4, 576) == gt
    print("GT equals the unsliced call's")
