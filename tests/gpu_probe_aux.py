"""Ad-hoc GPU probe (not a test): the aux configs of bench.py a few times each, for rocprofv3 --kernel-trace --stats."""
import hashlib, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
what = sys.argv[1:] or ["msm", "fav", "b4096"]
cache = m.BatchedBLSVerifierCache.init(max_sets=65536)
msg = hashlib.sha256(b"Mr F was here").digest()
rnd = msg
if "msm" in what:
    nm = 1 << int(os.environ.get("MSM_LOG2", "20"))
    rng = random.Random(7)
    base = bench.sign_records(m, cache, dev, range(2048), sks=[rng.getrandbits(96) | 1 for _ in range(2048)], msgs=[msg] * 2048)
    dp = base.view(2048, 320)[:, :96].contiguous().repeat(nm // 2048, 1).reshape(-1)
    sc = np.random.default_rng(7).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    for _ in range(6):
        m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    print("msm", cache.timings())
    # two MSMs in flight (two contexts, two streams, results left in device memory): what a caller that pipelines its calls gets
    import time
    c2 = [m.BatchedBLSVerifierCache.init(max_sets=64) for _ in range(2)]
    s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [torch.zeros(144, dtype=torch.uint8, device=dev) for _ in range(2)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            m.p1s_mult_pippenger_partial_device(c2[i % 2], outs[i % 2].data_ptr(), dp.data_ptr(), nm, ds.data_ptr(), 255, s2[i % 2].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print("msm two in flight: %.3f ms per MSM, %.1f M points/s" % (dt * 1e3, nm / dt / 1e6))
if "fav" in what:
    n = 32768
    sks = [bench.secret_key((1 << 41) + i) for i in range(n)]
    d_pks = bench.sign_records(m, cache, dev, range(n), sks=sks, msgs=[msg] * n).view(n, 320)[:, :96].contiguous()
    sig = bytes(bench.sign_records(m, cache, dev, [0], sks=[sum(sks) % bench.R_ORDER], msgs=[msg]).cpu().numpy())[128:320]
    for _ in range(6):
        assert m._check(m.lib().mi355_bls_fast_aggregate_verify_device(cache._h, d_pks.data_ptr(), n, msg, len(msg), sig, 0)) == 1
    print("fav", cache.timings())
if "b4096" in what:
    d4 = bench.sign_records(m, cache, dev, range(4096))
    c4 = m.BatchedBLSVerifierCache.init(max_sets=4096)
    for _ in range(6):
        assert c4.verify_device(d4.data_ptr(), 4096, rnd)
    print("b4096", c4.timings())
