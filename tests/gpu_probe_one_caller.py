import hashlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
d = bench.sign_records(m, gen, dev, range(65536))
c = m.BatchedBLSVerifierCache.init(max_sets=65536)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); assert c.verify_device(d.data_ptr(), 65536, rnd); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("MI355_BLS_FORKSIG_TEAM"), "one caller 65536: min %.3f median %.3f" % (min(ts), sorted(ts)[len(ts) // 2]), {k: round(v, 2) for k, v in c.timings().items()})
