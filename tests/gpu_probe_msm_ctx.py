"""Ad-hoc GPU probe (not a test): does the 2^20-point MSM of a context depend on what the process did before (streams, other contexts)?"""
import hashlib, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
msg = hashlib.sha256(b"Mr F was here").digest(); rnd = msg
cache = m.BatchedBLSVerifierCache.init(max_sets=65536)
nm = 1 << 20
rng = random.Random(7)
base = bench.sign_records(m, cache, dev, range(2048), sks=[rng.getrandbits(96) | 1 for _ in range(2048)], msgs=[msg] * 2048)
dp = base.view(2048, 320)[:, :96].contiguous().repeat(nm // 2048, 1).reshape(-1)
ds = torch.frombuffer(bytearray(np.random.default_rng(7).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()), dtype=torch.uint8).to(dev)
def msm(tag):
    m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    t0 = time.perf_counter()
    for _ in range(5):
        m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    dt = (time.perf_counter() - t0) / 5
    t = cache.timings()
    print("%-60s %.3f ms  sort %.2f buckets %.2f segments %.2f windows %.2f" % (tag, dt * 1e3, t["blinding"], t["hash_to_g2"], t["pk_mul"], t["sig_mul_sum"]), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "A"
if which == "A":
    msm("A fresh context")
d4 = bench.sign_records(m, cache, dev, range(4096))
if which in "BCD":
    assert cache.verify_device(d4.data_ptr(), 64, rnd)
    if which == "B":
        msm("B after a forked 64-set call on the same context")
if which in "CD":
    s4 = [torch.cuda.Stream(device=dev) for _ in range(16)]
    c4 = [m.BatchedBLSVerifierCache.init(max_sets=4096) for _ in range(16)]
    for c in c4:
        c.set_cooperative(which == "D")
    for c, st in zip(c4, s4):
        assert c.verify_device(d4.data_ptr(), 4096, rnd, st.cuda_stream)
    msm("%s + 16 streams and %s contexts used once" % (which, "latency-mode" if which == "D" else "throughput-mode"))
    for c in c4:
        c.close()
    msm("%s after closing them" % which)
