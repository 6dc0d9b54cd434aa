"""GPU parity: G1 Pippenger MSM (blst_p1s_mult_pippenger shape) through the C ABI.  The reference has no
expected-value test for MSM (parity unpinned by the reference); the oracles pin it."""
import random

import pytest

import bls12381_py as o
from util import g1_jac_to_affine, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="module")
def cache(m):
    return m.BatchedBLSVerifierCache.init(max_sets=64)


def test_msm_golden(m, cache):
    for v in golden("msm")["msm"]:
        out = m.p1s_mult_pippenger(cache, bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"]), v["nbits"])
        assert o.g1_to_blst_affine(g1_jac_to_affine(out)).hex() == v["result_affine"], v["n"]


def test_msm_edges(m, cache):
    v = golden("msm")["msm"][3]          # n = 32
    pts, sc = bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"])
    n = v["n"]
    P = [o.g1_from_blst_affine(pts[96 * i:96 * i + 96]) for i in range(n)]
    K = [int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(n)]
    assert m.p1s_mult_pippenger(cache, b"", b"", 255) == bytes(144)                      # empty -> infinity
    assert g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, bytes(32 * n), 255)) is None  # all-zero scalars
    for nbits in (1, 8, 64, 200, 256):
        got = g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, sc, nbits))
        assert got == o.msm_g1(P, K, nbits), nbits
    # repeated points / all-ones scalars / a point at infinity (all-zero affine) in the list
    pts2 = pts[:96] * 5 + bytes(96) + pts[96:192]
    sc2 = b"\xff" * (32 * 7)
    K2 = [(1 << 256) - 1] * 7
    P2 = [P[0]] * 5 + [None, P[1]]
    assert g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts2, sc2, 255)) == o.msm_g1(P2, K2, 255)
    # points ON THE CURVE BUT OUTSIDE G1 (blst's Pippenger does not ask for subgroup membership, and nothing here may assume it:
    # no endomorphism decomposition): x -> y = sqrt(x^3 + 4), order a multiple of the cofactor
    rng = random.Random(77)
    outside = []
    while len(outside) < 9:
        x = rng.randrange(o.P)
        rhs = (x * x * x + 4) % o.P
        y = pow(rhs, (o.P + 1) // 4, o.P)
        if y * y % o.P == rhs and o.g1_mul((x, y), o.R) is not None:
            outside.append((x, y))
    ks = [rng.getrandbits(255) for _ in outside]
    raw = b"".join(o.g1_to_blst_affine(q) for q in outside)
    got = g1_jac_to_affine(m.p1s_mult_pippenger(cache, raw, b"".join(k.to_bytes(32, "little") for k in ks), 255))
    assert got == o.msm_g1(outside, ks, 255)
    # cancellation: k*P + k*(-P) = inf
    neg = o.g1_to_blst_affine(o.g1_neg(P[0]))
    assert g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts[:96] + neg, sc[:32] * 2, 255)) is None


def test_msm_tail_meets_equal_and_opposite_operands(m, cache):
    """The Horner walk of the row tail (k_pip_rowtail: acc = 2^c acc + R_w) with inputs that make its addition exceptional, whatever the window
    width c the plan picks: [2^c] P + 2^c P has acc == R_0 (the doubling branch), [2^c] P - 2^c P has acc == -R_0 (infinity), and a scalar that is a
    multiple of every candidate 2^c leaves R_0 at infinity."""
    rng = random.Random(91)
    P = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
    le = lambda k: k.to_bytes(32, "little")
    for c in range(3, 18):
        Q = o.g1_mul(P, 1 << c)
        for pts, ks in (([P, Q], [1 << c, 1]), ([P, o.g1_neg(Q)], [1 << c, 1]), ([P, Q], [1 << c, 0]), ([P, Q, P], [1 << c, 1, (1 << c) + 1])):
            raw = b"".join(o.g1_to_blst_affine(q) for q in pts)
            for nbits in (c + 1, 64, 255):
                got = g1_jac_to_affine(m.p1s_mult_pippenger(cache, raw, b"".join(map(le, ks)), nbits))
                assert got == o.msm_g1(pts, ks, nbits), (c, ks, nbits)


@pytest.mark.parametrize("n", [1000, 20000, 50000])
def test_msm_vs_c_oracle(m, cache, n):
    """bench shape: P_i = [a_i]G with 96-bit a_i, 32 random scalar bytes, nbits = 255."""
    import c_oracle as co
    rng = random.Random(n)
    base = [co.sk_to_pk(rng.getrandbits(96) | 1) for _ in range(min(n, 2000))]
    pts = b"".join(base[i % len(base)] for i in range(n))
    sc = bytes(rng.getrandbits(8) for _ in range(32 * n))
    out = m.p1s_mult_pippenger(cache, pts, sc, 255)
    print("msm", n, "timings(ms):", cache.timings())
    assert o.g1_to_blst_affine(g1_jac_to_affine(out)) == co.msm_g1(pts, sc, 255)
    assert cache.timings()["total"] < 40.0          # full-width random scalars must not pile up in one bucket (top-window carry)


@pytest.mark.parametrize("n,nbits", [(40000, 255), (33000, 64), (70001, 130), (131072, 64), (100000, 256)])
def test_msm_lds_sort_path(m, cache, n, nbits):
    """n >= 2^15: the counting sort with a window's counters in LDS (k_pip_hist_lds / k_pip_scatter_lds) and the two window
    groups; 32-byte scalar images (two 16-byte loads), blst's own (nbits + 7) / 8 spacing (8: word loads, 17: byte loads), a
    ragged last slice, the exact blst argument list."""
    import c_oracle as co
    rng = random.Random(n)
    base = [co.sk_to_pk(rng.getrandbits(96) | 1) for _ in range(1500)]
    pts = b"".join(base[i % len(base)] for i in range(n))
    sb = (nbits + 7) // 8
    ks = [rng.getrandbits(nbits) for _ in range(n)]
    want = co.msm_g1(pts, b"".join(k.to_bytes(32, "little") for k in ks), nbits)
    got = m.blst_p1s_mult_pippenger(pts, b"".join(k.to_bytes(sb, "little") for k in ks), nbits)
    assert o.g1_to_blst_affine(g1_jac_to_affine(got)) == want
    if nbits == 255:
        got = m.p1s_mult_pippenger(cache, pts, b"".join(k.to_bytes(32, "little") for k in ks), nbits)
        assert o.g1_to_blst_affine(g1_jac_to_affine(got)) == want


def test_msm_linearity_at_2_20(m, cache):
    """Config 4 size (2^20 points): size-independent property MSM(k) + MSM(k') == MSM(k + k') on the same
    points (scalars chosen < 2^254 so the sum stays below 2^255), and MSM(0) is infinity."""
    import c_oracle as co
    import numpy as np
    rng = random.Random(99)
    n = 1 << 20
    base = [co.sk_to_pk(rng.getrandbits(96) | 1) for _ in range(4096)]
    pts = b"".join(base[i % 4096] for i in range(n))
    ra = np.random.default_rng(1)
    k1 = ra.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k2 = ra.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k1[:, 31] &= 0x3f
    k2[:, 31] &= 0x3f
    # 256-bit little-endian addition with numpy (column-wise carry)
    s = np.zeros((n, 32), dtype=np.uint8)
    carry = np.zeros(n, dtype=np.uint16)
    for j in range(32):
        t = k1[:, j].astype(np.uint16) + k2[:, j].astype(np.uint16) + carry
        s[:, j] = (t & 0xff).astype(np.uint8)
        carry = t >> 8
    a = g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, k1.tobytes(), 255))
    print("msm 2^20 timings(ms):", cache.timings())
    b = g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, k2.tobytes(), 255))
    c = g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, s.tobytes(), 255))
    assert o.g1_add(a, b) == c and c is not None


def _bench_points(n, seed, distinct=4096):
    import c_oracle as co
    rng = random.Random(seed)
    base = [co.sk_to_pk(rng.getrandbits(96) | 1) for _ in range(distinct)]
    return b"".join(base[i % distinct] for i in range(n))


def test_msm_config4_vs_c_oracle_at_2_20(m, cache):
    """BASELINE config 4 at its own size: 2^20 points x 255-bit scalars (16 windows of 16 bits, 2^15 buckets, 32 sort slices, two
    window groups on two streams) against the C restatement of the bucket method (oracle_msm_g1_pippenger, OpenMP over the
    windows), through the host-array entry and the device-resident one."""
    import c_oracle as co
    import numpy as np
    import torch
    n = 1 << 20
    pts = _bench_points(n, 2020)
    sc = np.random.default_rng(2020).integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
    want = co.msm_g1_pippenger(pts, sc, 255)
    got = m.p1s_mult_pippenger(cache, pts, sc, 255)
    assert o.g1_to_blst_affine(g1_jac_to_affine(got)) == want
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    got = m.p1s_mult_pippenger_device(cache, dp.data_ptr(), n, ds.data_ptr(), 255)
    print("msm 2^20 timings(ms):", cache.timings())
    assert o.g1_to_blst_affine(g1_jac_to_affine(got)) == want


@pytest.mark.parametrize("ngpu", [1, 2, 3, 8])
def test_msm_multi_equals_c_oracle_at_2_17(m, ngpu):
    """Point-sharded MSM over ngpu contexts (several contexts on ONE device stand in for several devices): 2^17 + 5 points (ragged
    shards) x 255 bits == the C restatement; empty input and fewer points than devices."""
    import c_oracle as co
    import numpy as np
    n = (1 << 17) + 5
    pts = _bench_points(n, 17)
    sc = np.random.default_rng(17).integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
    want = co.msm_g1_pippenger(pts, sc, 255)
    caches = [m.BatchedBLSVerifierCache.init(max_sets=64) for _ in range(ngpu)]
    got = m.p1s_mult_pippenger_multi(caches, pts, sc, 255)
    assert o.g1_to_blst_affine(g1_jac_to_affine(got)) == want
    assert m.p1s_mult_pippenger_multi(caches, b"", b"", 255) == bytes(144)
    small = m.p1s_mult_pippenger_multi(caches, pts[:96 * 3], sc[:32 * 3], 255)           # 3 points: some devices get none
    assert o.g1_to_blst_affine(g1_jac_to_affine(small)) == co.msm_g1(pts[:96 * 3], sc[:32 * 3], 255)
    covered = 0
    for g in range(ngpu):
        first, count = m.msm_shard_range(n, ngpu, g)
        assert first == covered
        covered += count
    assert covered == n
    for c in caches:
        c.close()


def test_msm_multi_at_2_20_and_partial_merge(m, cache):
    """2^20 points: 8 shards == the single-device result == the C restatement; the one-process-per-GPU form (every rank's partial
    left in device memory, p1s_add_device on the gathered partials) and the host merge (p1s_add) agree; G2 shards likewise."""
    import c_oracle as co
    import numpy as np
    import torch
    from util import g2_jac_to_affine
    n = 1 << 20
    pts = _bench_points(n, 2020)
    sc = np.random.default_rng(2020).integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
    single = g1_jac_to_affine(m.p1s_mult_pippenger(cache, pts, sc, 255))
    assert o.g1_to_blst_affine(single) == co.msm_g1_pippenger(pts, sc, 255)
    caches = [m.BatchedBLSVerifierCache.init(max_sets=64) for _ in range(8)]
    assert g1_jac_to_affine(m.p1s_mult_pippenger_multi(caches, pts, sc, 255)) == single
    # device-resident shards + the rank-style flow
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    ranges = [m.msm_shard_range(n, 8, g) for g in range(8)]
    got = m.p1s_mult_pippenger_multi_device(caches, [dp.data_ptr() + 96 * f for f, c in ranges], n, [ds.data_ptr() + 32 * f for f, c in ranges], 255)
    assert g1_jac_to_affine(got) == single
    gathered = torch.zeros(8 * 256, dtype=torch.uint8, device="cuda")                   # 8 partials, 256 bytes apart
    for g, (f, c) in enumerate(ranges):
        m.p1s_mult_pippenger_partial_device(caches[g], gathered.data_ptr() + 256 * g, dp.data_ptr() + 96 * f, c, ds.data_ptr() + 32 * f, 255)
    torch.cuda.synchronize()
    assert g1_jac_to_affine(m.p1s_add_device(caches[0], gathered.data_ptr(), 8, 256)) == single
    parts = bytes(gathered.cpu().numpy())
    assert g1_jac_to_affine(m.p1s_add(caches[0], [parts[256 * g:256 * g + 144] for g in range(8)])) == single
    assert g1_jac_to_affine(m.p1s_add(caches[0], [bytes(144), parts[:144], bytes(144)])) == g1_jac_to_affine(parts[:144])     # infinity partials
    # G2: 3 shards of 5000 points x 64-bit scalars (combine's shape)
    rng = random.Random(5)
    h = co.hash_to_g2(b"g2 multi", o.DST_SIG)
    base = [co.g2_mul(h, rng.randrange(1, o.R)) for _ in range(64)]
    q = b"".join(base[i % 64] for i in range(5000))
    k = b"".join(rng.getrandbits(64).to_bytes(32, "little") for _ in range(5000))
    got2 = m.p1s_mult_pippenger_multi(caches[:3], q, k, 64, g2=True)
    assert o.g2_to_blst_affine(g2_jac_to_affine(got2)) == co.msm_g2(q, k, 64, 32)
    for c in caches:
        c.close()
