"""Ad-hoc GPU probe (not a test): per-stage HIP-event timings of the latency-bound calls."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
msg = hashlib.sha256(b"Mr F was here").digest()
n = 32768
sks = [bench.secret_key((1 << 41) + i) for i in range(n)]
d_pks = bench.sign_records(m, gen, dev, range(n), sks=sks, msgs=[msg] * n).view(n, 320)[:, :96].contiguous()
sig = bytes(bench.sign_records(m, gen, dev, [0], sks=[sum(sks) % bench.R_ORDER], msgs=[msg]).cpu().numpy())[128:320]
for k in (32768, 1):
    f = lambda: m._check(m.lib().mi355_bls_fast_aggregate_verify_device(gen._h, d_pks.data_ptr(), k, msg, len(msg), sig, 0))
    f(); t0 = time.perf_counter(); r = f(); dt = (time.perf_counter() - t0) * 1e3
    t = gen.timings()
    print("FAV n=%d: %.2f ms verdict %d  g1sum %.2f hash+setup %.2f lines %.2f products %.2f tail %.2f" % (k, dt, r, t["blinding"], t["hash_to_g2"], t["pk_mul"], t["sig_mul_sum"], t["miller_lines"]))
base = bench.sign_records(m, gen, dev, range(65536))
SIZES = tuple(int(x) for x in os.environ["LAT_SIZES"].split(",")) if os.environ.get("LAT_SIZES") else (3, 64, 1000, 4096, 8192, 16384, 65536)
for nn in SIZES:
    c = m.BatchedBLSVerifierCache.init(max_sets=nn)
    c.verify_device(base.data_ptr(), nn, rnd); c.verify_device(base.data_ptr(), nn, rnd)
    t0 = time.perf_counter(); ok = c.verify_device(base.data_ptr(), nn, rnd); dt = (time.perf_counter() - t0) * 1e3
    print("batch n=%d: %.2f ms %s %s %s" % (nn, dt, ok, {k: round(v, 2) for k, v in c.timings().items()}, {k: round(v, 2) for k, v in c.kernel_timings().items()}))
    c.close()
