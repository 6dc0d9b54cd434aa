"""Ad-hoc GPU probe: MSM 2^20 timing (not a test)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import numpy as np
m = ge.load_package()
import c_oracle as co
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = random.Random(1)
base = [co.sk_to_pk(rng.getrandbits(96) | 1) for _ in range(4096)]
pts = b"".join(base[i % 4096] for i in range(n))
sc = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
cache = m.BatchedBLSVerifierCache.init(max_sets=64)
dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
for it in range(3):
    t0 = time.time(); out = m.p1s_mult_pippenger_device(cache, dp.data_ptr(), n, ds.data_ptr(), 255); dt = time.time() - t0
    t = cache.timings()
    print(n, "wall %.2f ms" % (dt * 1e3), "sort %.2f bucket %.2f segred %.2f winsum %.2f total %.2f" % (t["blinding"], t["hash_to_g2"], t["pk_mul"], t["sig_mul_sum"], t["total"]), flush=True)
