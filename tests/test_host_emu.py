"""CPU execution (tests/host_emu) of the product's device arithmetic against the golden fixtures
and the oracle: field ops, SHA-256, hash_to_field, SSWU, isogeny, hash_to_G2, scalar mults, pairing."""
import ctypes
import hashlib
import random

import bls12381_py as o
from util import (buf, fp12_from_bytes, fp12_hexlist_to_flat, g1_aff_to_jac_bytes, g1_jac_to_affine,
                  g2_aff_to_jac_bytes, g2_jac_to_affine, golden)


def call(emu, fn, *ins, outlen=48):
    out = buf(outlen)
    getattr(emu, fn)(*ins, out)
    return out.raw


def test_fields(emu):
    d = golden("fields")
    for a, b, m, s, df in d["fp_mul"]:
        A, B = bytes.fromhex(a), bytes.fromhex(b)
        assert call(emu, "emu_fp_mul", A, B).hex() == m
        assert call(emu, "emu_fp_add", A, B).hex() == s
        assert call(emu, "emu_fp_sub", A, B).hex() == df
    for a, i in d["fp_inv"]:
        assert call(emu, "emu_fp_inv", bytes.fromhex(a)).hex() == i
    for a, b, c, s, i in d["fp2_mul"]:
        A, B = bytes.fromhex("".join(a)), bytes.fromhex("".join(b))
        assert call(emu, "emu_fp2_mul", A, B, outlen=96).hex() == "".join(c)
        assert call(emu, "emu_fp2_sqr", A, outlen=96).hex() == "".join(s)
        assert call(emu, "emu_fp2_inv", A, outlen=96).hex() == "".join(i)


def test_sha256(emu):
    rng = random.Random(1)
    for n in [0, 1, 31, 32, 55, 56, 63, 64, 65, 119, 120, 128, 200]:
        m = bytes(rng.randrange(256) for _ in range(n))
        assert call(emu, "emu_sha256", m, n, outlen=32) == hashlib.sha256(m).digest()


def test_hash_to_g2_stages(emu):
    for v in golden("h2c"):
        m = bytes.fromhex(v["msg"])
        dst = v["dst"].encode()
        u = call(emu, "emu_hash_to_field", m, len(m), dst, len(dst), outlen=192)
        assert u.hex() == "".join(v["u"][0]) + "".join(v["u"][1])
        for j, key in ((0, "q0"), (1, "q1")):
            q = call(emu, "emu_sswu", u[96 * j:96 * j + 96], outlen=288)
            assert g2_jac_to_affine(q) == o.g2_from_blst_affine(bytes.fromhex(v[key]))
            qi = call(emu, "emu_iso3", q, outlen=288)
            assert g2_jac_to_affine(qi) == o.iso3_g2(o.g2_from_blst_affine(bytes.fromhex(v[key])))
        h = call(emu, "emu_hash_to_g2", m, len(m), dst, len(dst), outlen=288)
        ha = g2_jac_to_affine(h)
        assert o.g2_to_blst_affine(ha).hex() == v["h"]
        assert o.g2_compress(ha).hex() == v["h_compressed"]


def test_scalar_mul_and_add(emu):
    rng = random.Random(5)
    p = o.g1_mul(o.G1_GEN, rng.randrange(o.R))
    q = o.g2_mul(o.G2_GEN, rng.randrange(o.R))
    for kk in [1, 2, 3, 0xffffffffffffffff, rng.getrandbits(64), 1 << 63]:
        r1 = call(emu, "emu_g1_mul_u64", o.g1_to_blst_affine(p), ctypes.c_uint64(kk), outlen=144)
        assert g1_jac_to_affine(r1) == o.g1_mul(p, kk)
        r2 = call(emu, "emu_g2_mul_u64", o.g2_to_blst_affine(q), ctypes.c_uint64(kk), outlen=288)
        assert g2_jac_to_affine(r2) == o.g2_mul(q, kk)
    # complete addition: P+P, P+(-P), inf+P, P+inf
    pj, qj = g1_aff_to_jac_bytes(p), g2_aff_to_jac_bytes(q)
    assert g1_jac_to_affine(call(emu, "emu_g1_add", pj, pj, outlen=144)) == o.g1_add(p, p)
    assert g1_jac_to_affine(call(emu, "emu_g1_add", pj, g1_aff_to_jac_bytes(o.g1_neg(p)), outlen=144)) is None
    assert g1_jac_to_affine(call(emu, "emu_g1_add", bytes(144), pj, outlen=144)) == p
    assert g2_jac_to_affine(call(emu, "emu_g2_add", qj, qj, outlen=288)) == o.g2_add(q, q)
    assert g2_jac_to_affine(call(emu, "emu_g2_add", qj, bytes(288), outlen=288)) == q


def test_pairing_golden(emu):
    for v in golden("pairing")["vectors"]:
        pj = bytes.fromhex(v["p"]) + o.fp_to_mont_bytes(1)
        qj = bytes.fromhex(v["q"]) + o.fp_to_mont_bytes(1) + bytes(48)
        out = buf(576)
        emu.emu_pairing_product(pj, qj, 1, out, 1)
        assert fp12_from_bytes(out.raw) == fp12_hexlist_to_flat(v["gt3"])


def test_pairing_projective_inputs_and_product(emu):
    """Jacobian (non-normalised) P and Q, and a 2-pair product that must be 1: e(aG,Q) e(-G, aQ)."""
    rng = random.Random(9)
    a = rng.randrange(1, o.R)
    q = o.g2_mul(o.G2_GEN, rng.randrange(1, o.R))
    p1 = o.g1_mul(o.G1_GEN, a)
    q2 = o.g2_mul(q, a)
    zs = rng.randrange(2, o.P)
    # scale P to Jacobian with z = zs
    pj = o.fp_to_mont_bytes(p1[0] * zs * zs % o.P) + o.fp_to_mont_bytes(p1[1] * pow(zs, 3, o.P) % o.P) + o.fp_to_mont_bytes(zs)
    z2 = (rng.randrange(o.P), rng.randrange(o.P))
    z22 = o.f2sqr(z2)
    qx = o.f2mul(q[0], z22)
    qy = o.f2mul(q[1], o.f2mul(z22, z2))
    qj = b"".join(o.fp_to_mont_bytes(c) for c in (qx[0], qx[1], qy[0], qy[1], z2[0], z2[1]))
    out = buf(576)
    emu.emu_pairing_product(pj, qj, 1, out, 1)
    assert fp12_from_bytes(out.raw) == o.pairing(p1, q)
    ps = pj + g1_aff_to_jac_bytes(o.g1_neg(o.G1_GEN))
    qs = qj + g2_aff_to_jac_bytes(q2)
    emu.emu_pairing_product(ps, qs, 2, out, 1)
    assert fp12_from_bytes(out.raw) == o.F12_ONE
    # infinity pair contributes 1
    ps3 = ps + bytes(144)
    qs3 = qs + g2_aff_to_jac_bytes(q)
    emu.emu_pairing_product(ps3, qs3, 3, out, 1)
    assert fp12_from_bytes(out.raw) == o.F12_ONE
