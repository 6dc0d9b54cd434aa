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
