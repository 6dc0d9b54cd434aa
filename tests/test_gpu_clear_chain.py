"""k_hash_clear (round 5: one hand-allocated assembly statement, tools/gen_clear_asm.py) and the lane-team engine's cofactor clearing (round 6:
k_team_clear + k_clear_fix, csrc/teamvm.hpp; a latency-mode context takes it) on the device, through the test hook
mi355_bls_debug_g2_clear_cofactor: H = clear_cofactor(q0 + q1) for pairs of arbitrary points of E2 against the oracle - including the inputs
no hash produces and the loop does NOT handle itself (it flags them and the kernel recomputes the lane with the complete formulas): points at
infinity, q0 == q1, q0 == -q1.  Reference: the cofactor clearing of hash-to-G2 (blst_abi.nim:383, RFC 9380 G.3)."""
import ctypes
import random

import pytest

import bls12381_py as o
from util import g2_jac_to_affine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _jac_bytes(p, z):
    """affine p -> blst_p2 image with the Jacobian coordinate Z = z (an Fp2 element): (x z^2, y z^3, z)"""
    if p is None:
        return bytes(288)
    z2 = o.f2sqr(z)
    x, y = o.f2mul(p[0], z2), o.f2mul(p[1], o.f2mul(z2, z))
    return b"".join(o.fp_to_mont_bytes(c) for c in (x[0], x[1], y[0], y[1], z[0], z[1]))


@pytest.mark.parametrize("mode", ["throughput", "latency", "latency_rows2", "latency_team"])
def test_clear_cofactor_of_arbitrary_pairs(m, mode):
    rng = random.Random(3)

    def e2_point():          # a point of E2(Fp2), generally outside G2: what the SSWU map + isogeny produce
        return o.iso3_g2(o.sswu_g2((rng.randrange(o.P), rng.randrange(o.P))))

    def z():
        return (rng.randrange(1, o.P), rng.randrange(o.P))

    pairs = [(e2_point(), e2_point()) for _ in range(70)]
    a, b = e2_point(), e2_point()
    pairs += [(a, None), (None, b), (None, None), (a, a), (a, o.g2_neg(a)), (o.G2_GEN, o.g2_mul(o.G2_GEN, 2))]
    if mode == "latency_rows2":                     # 304 pairs: the row executor at two workgroups per CU
        pairs = pairs * 4
    if mode == "latency_team":                      # 456 pairs, beyond what the row executor takes (four waves per message): the lane-team engine, a message per 16 lanes
        pairs = pairs * 6
    blob = b"".join(_jac_bytes(p, z()) + _jac_bytes(q, z()) for p, q in pairs)
    out = ctypes.create_string_buffer(288 * len(pairs))
    cache = m.BatchedBLSVerifierCache.init(max_sets=1024)
    cache.set_cooperative(mode != "throughput")     # latency: the engine's programs on rows (k_team_clear_rows) or lane teams (k_team_clear); throughput: k_hash_clear
    assert m._check(m.lib().mi355_bls_debug_g2_clear_cofactor(cache._h, blob, len(pairs), out)) == 0
    for i, (p, q) in enumerate(pairs):
        got = g2_jac_to_affine(out.raw[288 * i:288 * i + 288])
        s = o.g2_add(p, q)
        want = None if s is None else o.clear_cofactor_g2(s)
        assert got == want, i
        assert got is None or o.g2_in_subgroup(got)
