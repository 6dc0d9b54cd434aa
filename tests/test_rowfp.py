"""The limb-parallel row arithmetic of the MSM's tail (nim-blscurve_amd/csrc/rowfp.hpp: one Fp value along the 16 lanes of a DPP row, four rows per wave)
executed on 64 emulated lanes (tests/host_emu, every accumulator / carry / shift bound asserted) against big-integer arithmetic and the oracle's group law.
The GPU parity tests of the MSM (tests/test_gpu_msm.py, test_gpu_headline.py) run the kernel that uses it.  Reference: blst_abi.nim:336-340."""
import random

import bls12381_py as o
from util import buf, fp12_from_bytes, fp12_to_bytes, g1_aff_to_jac_bytes, g1_jac_to_affine


def _mont(v):
    return o.fp_to_mont_bytes(v % o.P)


def test_row_mul_is_the_montgomery_product(emu):
    rng = random.Random(5)
    cases = [[rng.randrange(o.P) for _ in range(4)] for _ in range(6)]
    cases += [[0, 1, o.P - 1, (o.P - 1) // 2], [o.P - 1] * 4, [1 << 380, (1 << 381) - 1, 3, o.P - 2]]
    for twice in (0, 1):
        for a in cases:
            b = [rng.randrange(o.P) for _ in range(4)] if a[0] else [o.P - 1, o.P - 1, o.P - 1, 0]
            out = buf(192)
            emu.emu_row_mul(b"".join(map(_mont, a)), b"".join(map(_mont, b)), out, twice)
            got = [o.fp_from_mont_bytes(out.raw[48 * i:48 * i + 48]) for i in range(4)]
            assert got == [(1 + twice) * x * y % o.P for x, y in zip(a, b)]


def _jac(p, z):
    return _mont(p[0] * z * z) + _mont(p[1] * z * z * z) + _mont(z)


def test_row_doubling_chain(emu):
    rng = random.Random(6)
    p = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
    for times in (1, 2, 17):
        out = buf(144)
        emu.emu_row_dbl(_jac(p, rng.randrange(1, o.P)), out, times)
        assert g1_jac_to_affine(out.raw) == o.g1_mul(p, 1 << times)
    out = buf(144)
    emu.emu_row_dbl(bytes(144), out, 3)                      # infinity stays infinity
    assert g1_jac_to_affine(out.raw) is None


def test_row_addition_and_its_exceptional_cases(emu):
    rng = random.Random(7)
    p = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
    q = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
    z1, z2 = rng.randrange(1, o.P), rng.randrange(1, o.P)
    for dbl_first in (0, 2):
        pp = o.g1_mul(p, 1 << dbl_first)
        for a, b, want in ((_jac(p, z1), _jac(q, z2), o.g1_add(pp, q)),
                           (_jac(p, z1), bytes(144), pp),                                      # Q at infinity
                           (bytes(144), _jac(q, z2), q),                                       # P at infinity
                           (bytes(144), bytes(144), None),
                           (_jac(p, z1), _jac(o.g1_neg(pp), z2), None),                        # P = -Q
                           (_jac(p, z1), _jac(pp, z2), o.g1_mul(pp, 2)),                       # P = Q under different Z: the doubling branch
                           (_jac(p, z1), g1_aff_to_jac_bytes(q), o.g1_add(pp, q))):
            out = buf(144)
            emu.emu_row_add(a, b, out, dbl_first)
            assert g1_jac_to_affine(out.raw) == want


def test_row_exponentiation_of_the_square_root(emu):
    """k_hash_one's two SSWU square-root chains run along rows: a^((p-3)/4) by the library's own sliding-window schedule"""
    rng = random.Random(8)
    for a in ([rng.randrange(1, o.P) for _ in range(4)], [1, o.P - 1, 2, (o.P - 1) // 2]):
        out = buf(192)
        emu.emu_row_pow(b"".join(map(_mont, a)), out)
        got = [o.fp_from_mont_bytes(out.raw[48 * i:48 * i + 48]) for i in range(4)]
        assert got == [pow(x, (o.P - 3) // 4, o.P) for x in a]


def test_cyclotomic_squarings_on_rows(emu):
    """the final exponentiation's squarings by |x| (csrc/rowcyc.hpp: Granger-Scott in the flat basis, eighteen products along rows) on unitary values"""
    rng = random.Random(9)
    for n in (1, 2, 9, 32):
        f = tuple((rng.randrange(o.P), rng.randrange(o.P)) for _ in range(6))
        t = o.f12mul(o.f12conj(f), o.f12inv(f))              # the easy part of the final exponentiation makes it unitary
        t = o.f12mul(o.f12frob_n(t, 2), t)
        out = buf(576)
        emu.emu_cyc_sqr(fp12_to_bytes(t), n, out)
        want = t
        for _ in range(n):
            want = o.f12sqr(want)
        assert fp12_from_bytes(out.raw) == want
    out = buf(576)
    emu.emu_cyc_sqr(fp12_to_bytes(o.F12_ONE), 5, out)
    assert fp12_from_bytes(out.raw) == o.F12_ONE
