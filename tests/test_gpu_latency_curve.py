"""One blocking batchVerify per size, the sizes of bench.py's `aux.latency_curve` (the reference's own benchmark uses batches of 6 / 60 /
180 signatures, benchmarks/bench_all.nim:48-65; 65 536 is the headline batch): every verdict true, a tampered batch false at every size,
and the throughput of a blocking call grows with the batch - the curve INTEGRATION.md's "When to call the GPU" is read from."""
import hashlib
import time

import pytest

pytestmark = pytest.mark.gpu

RND = hashlib.sha256(b"Mr F was here").digest()


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def test_latency_curve_is_monotone_in_throughput(m):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
    d = bench.sign_records(m, gen, dev, range(9_000_000, 9_000_000 + 65536))
    gen.close()
    bad = d.clone()
    bad[96] ^= 1                                         # tuple 0 signs another message
    rates = []
    for n in bench.LATENCY_CURVE_SIZES:
        c = m.BatchedBLSVerifierCache.init(max_sets=n)
        assert c.verify_device(d.data_ptr(), n, RND) is True
        assert c.verify_device(bad.data_ptr(), n, RND) is False
        best = None
        for _ in range(3):                                # the best of three: a blocking call of a few milliseconds is exposed to host jitter
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert c.verify_device(d.data_ptr(), n, RND) is True
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rates.append((n, n / best, best * 1e3))
        c.close()
    print("latency curve (n, verifications/s, ms):", [(n, round(r), round(ms, 3)) for n, r, ms in rates])
    for (n0, r0, _), (n1, r1, _) in zip(rates, rates[1:]):
        assert r1 > r0, ("throughput of a blocking call must grow with the batch", n0, r0, n1, r1)
    # the small end is latency-bound (a few milliseconds whatever the size), the large end is within reach of the pipelined rate
    assert rates[0][2] < 8.0 and rates[-1][1] > 3.0e6
