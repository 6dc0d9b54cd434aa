"""Ad-hoc GPU probe (not a test): does a small blocking batchVerify depend on the context's capacity or on how many streams exist in the process
(HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues; a caller stream and its fork streams that share one serialise)?"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
gen = m.BatchedBLSVerifierCache.init(max_sets=4096)
base = bench.sign_records(m, gen, dev, range(4096))
def run(c, n, stream=0):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); assert c.verify_device(base.data_ptr(), n, rnd, stream); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for cap in (64, 4096, 65536):
    c = m.BatchedBLSVerifierCache.init(max_sets=cap)
    print("cap %6d: n=64 %.3f ms  n=%d %.3f ms  %s" % (cap, run(c, 64), min(cap, 4096), run(c, min(cap, 4096)), {k: round(v, 2) for k, v in c.timings().items()}))
    c.close()
extra = [torch.cuda.Stream(device=dev) for _ in range(3)]
others = [m.BatchedBLSVerifierCache.init(max_sets=4096) for _ in range(3)]
for k in range(4):
    c = m.BatchedBLSVerifierCache.init(max_sets=4096)
    print("after %d more contexts: n=64 %.3f ms  n=4096 %.3f ms on the null stream; n=64 %.3f on a torch stream" % (3 + k, run(c, 64), run(c, 4096), run(c, 64, extra[k % 3].cuda_stream)))
    others.append(c)
