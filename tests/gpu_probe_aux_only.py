import hashlib, os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
cache = m.BatchedBLSVerifierCache.init(max_sets=65536, numThreads=4096, device=0)
out = bench.aux_rows(m, cache, dev)
a = out["g1_msm_2^20"]
print("aux_rows alone: msm %.3f ms two-in-flight %.3f" % (a["ms_per_call"], a["ms_per_msm_two_in_flight"]), {k: round(v, 2) for k, v in a["stage_ms"].items()})
