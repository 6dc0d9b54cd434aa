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


def test_inversion_by_division_steps_equals_fermat(emu):
    """fp_inv (Bernstein-Yang division steps, 62 per batch) against the exponentiation a^(p-2) it replaced, and a * 1/a = 1:
    random values, the edge values 0, 1, 2, p - 1, p - 2 and small / sparse ones (few batches, early exit)."""
    rng = random.Random(11)
    vals = [0, 1, 2, o.P - 1, o.P - 2, (o.P + 1) // 2, 1 << 380, (1 << 62) - 1, 1 << 62, (1 << 124) + 1]
    vals += [rng.randrange(o.P) for _ in range(300)] + [rng.getrandbits(b) for b in (1, 7, 33, 63, 64, 65, 190, 379)]
    two = (2).to_bytes(48, "little")
    one = call(emu, "emu_fp_mul", two, call(emu, "emu_fp_inv_fermat", two))          # the image of 1 in the byte format of the harness
    for v in vals:
        a = (v % o.P).to_bytes(48, "little")
        got = call(emu, "emu_fp_inv", a)
        assert got == call(emu, "emu_fp_inv_fermat", a), hex(v)
        if v % o.P:
            assert call(emu, "emu_fp_mul", a, got) == one


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


def test_rfc9380_vector_through_the_device_arithmetic(emu):
    """the RFC 9380 J.10.1 vector of the empty message (tests/test_oracle_kats.py says where it comes from) through the product's own hash_to_g2
    (the code the kernels run, here on the CPU under the bounds tracker): affine output byte for byte"""
    dst = b"QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_"
    h = g2_jac_to_affine(call(emu, "emu_hash_to_g2", b"", 0, dst, len(dst), outlen=288))
    assert h == ((int("0141ebfbdca40eb85b87142e130ab689c673cf60f1a3e98d69335266f30d9b8d4ac44c1038e9dcdd5393faf5c41fb78a", 16),
                  int("05cb8437535e20ecffaef7752baddf98034139c38452458baeefab379ba13dff5bf5dd71b72418717047f5b0f37da03d", 16)),
                 (int("0503921d7f6a12805e72940b963c0cf3471c7b2a524950ca195d11062ee75ec076daf2d4bc358c4b190c0c98064fdd92", 16),
                  int("12424ac32561493f3fe3c260708a12b7c620e7be00099a974e259ddc7d1f6395c3c811cdd19f1e8dbf3e9ecfdcbab8d6", 16)))


def test_hash_to_field_fast_path_for_32_byte_messages(emu):
    """k_hash_map's specialised expand_message_xmd (message-independent words precomputed per DST) against the
    generic byte-wise one and the oracle, for several DST lengths incl. the edges of its range."""
    rng = random.Random(5)
    for dst in [o.DST_SIG, b"BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_", b"x" * 28, b"y" * 83, b"QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_"]:
        for _ in range(3):
            msg = bytes(rng.getrandbits(8) for _ in range(32))
            a, b = buf(192), buf(192)
            assert emu.emu_hash_to_field_msg32(msg, dst, len(dst), a) == 1
            emu.emu_hash_to_field(msg, 32, dst, len(dst), b)
            assert a.raw == b.raw
            u = o.hash_to_field_fp2(msg, dst, 2)
            assert a.raw == b"".join(o.fp_to_mont_bytes(c) for c in (u[0][0], u[0][1], u[1][0], u[1][1]))
    out = buf(192)
    assert emu.emu_hash_to_field_msg32(bytes(32), b"z" * 27, 27, out) == 0 and emu.emu_hash_to_field_msg32(bytes(32), b"z" * 84, 84, out) == 0


def test_scalar_mul_and_add(emu):
    rng = random.Random(5)
    p = o.g1_mul(o.G1_GEN, rng.randrange(o.R))
    q = o.g2_mul(o.G2_GEN, rng.randrange(o.R))
    for kk in [1, 2, 3, 0xffffffffffffffff, rng.getrandbits(64), 1 << 63]:
        r1 = call(emu, "emu_g1_mul_u64", o.g1_to_blst_affine(p), ctypes.c_uint64(kk), outlen=144)
        assert g1_jac_to_affine(r1) == o.g1_mul(p, kk)
        r2 = call(emu, "emu_g2_mul_u64", o.g2_to_blst_affine(q), ctypes.c_uint64(kk), outlen=288)
        assert g2_jac_to_affine(r2) == o.g2_mul(q, kk)
    # signed 4-bit windows (k_pkmul): digit edge cases 8/9 (carry), all-ones, top carry, tiny scalars
    for kk in [1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x99, 0xffffffffffffffff, 0x8888888888888888, 0x9999999999999999, 0x7fffffffffffffff,
               1 << 63, rng.getrandbits(64), rng.getrandbits(64), rng.getrandbits(64)]:
        r1 = call(emu, "emu_g1_mul_u64_w4", o.g1_to_blst_affine(p), ctypes.c_uint64(kk), outlen=144)
        assert g1_jac_to_affine(r1) == o.g1_mul(p, kk), hex(kk)
    for kk in [1, 9, 0xffffffffffffffff, rng.getrandbits(64)]:
        r2 = call(emu, "emu_g2_mul_u64_w4", o.g2_to_blst_affine(q), ctypes.c_uint64(kk), outlen=288)
        assert g2_jac_to_affine(r2) == o.g2_mul(q, kk), hex(kk)
    # 256-bit scalars (batch signer): affine base in G1, Jacobian base in G2
    for kk in [1, o.R - 1, rng.randrange(o.R), (1 << 255) + 12345]:
        k32 = kk.to_bytes(32, "little")
        r1 = call(emu, "emu_g1_mul_256", o.g1_to_blst_affine(p), k32, outlen=144)
        assert g1_jac_to_affine(r1) == o.g1_mul(p, kk)
        r2 = call(emu, "emu_g2_mul_256_jac", g2_aff_to_jac_bytes(q), k32, outlen=288)
        assert g2_jac_to_affine(r2) == o.g2_mul(q, kk)
    # complete addition: P+P, P+(-P), inf+P, P+inf
    pj, qj = g1_aff_to_jac_bytes(p), g2_aff_to_jac_bytes(q)
    assert g1_jac_to_affine(call(emu, "emu_g1_add", pj, pj, outlen=144)) == o.g1_add(p, p)
    assert g1_jac_to_affine(call(emu, "emu_g1_add", pj, g1_aff_to_jac_bytes(o.g1_neg(p)), outlen=144)) is None
    assert g1_jac_to_affine(call(emu, "emu_g1_add", bytes(144), pj, outlen=144)) == p
    assert g2_jac_to_affine(call(emu, "emu_g2_add", qj, qj, outlen=288)) == o.g2_add(q, q)
    assert g2_jac_to_affine(call(emu, "emu_g2_add", qj, bytes(288), outlen=288)) == q
    # extended-Jacobian (XYZZ) mixed addition of the Pippenger buckets: sums with every special case on the way - first
    # operand at infinity, an affine infinity, P + P (doubling from the affine operand), P + (-P) -> infinity -> + Q
    p2, p3 = o.g1_mul(p, 7), o.g1_mul(p, 1000003)
    a1 = o.g1_to_blst_affine
    for pts, want in [([p], p), ([p, p2, p3], o.g1_add(o.g1_add(p, p2), p3)), ([p, p], o.g1_add(p, p)),
                      ([p, p, p], o.g1_mul(p, 3)), ([p, o.g1_neg(p)], None), ([p, o.g1_neg(p), p2], p2),
                      ([None, p, None, p2], o.g1_add(p, p2)), ([None], None), ([p, p2, o.g1_add(p, p2)], o.g1_mul(o.g1_add(p, p2), 2))]:
        raw = b"".join(a1(x) if x is not None else bytes(96) for x in pts)
        assert g1_jac_to_affine(call(emu, "emu_g1_sum_xyzz", raw, ctypes.c_uint32(len(pts)), outlen=144)) == want
    q2 = o.g2_mul(q, 11)
    a2 = o.g2_to_blst_affine
    for pts, want in [([q, q2], o.g2_add(q, q2)), ([q, q], o.g2_add(q, q)), ([q, o.g2_neg(q), q2], q2), ([None, q, None], q),
                      ([q, q2, o.g2_add(q, q2)], o.g2_mul(o.g2_add(q, q2), 2))]:
        raw = b"".join(a2(x) if x is not None else bytes(192) for x in pts)
        assert g2_jac_to_affine(call(emu, "emu_g2_sum_xyzz", raw, ctypes.c_uint32(len(pts)), outlen=288)) == want


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
    # longer per-step line products (the accumulator is only carried between lines, not reduced: the bound
    # tracker of this build checks that this stays within the multiplier's preconditions)
    emu.emu_pairing_product(ps3 * 3, qs3 * 3, 9, out, 1)
    assert fp12_from_bytes(out.raw) == o.F12_ONE


def _random_curve_point_g1(rng):
    """A random point of E1(Fp) (almost surely NOT in G1: cofactor 0x396c8c005555e1568c00aaab0000aaab)."""
    while True:
        x = rng.randrange(o.P)
        y = o.fp_sqrt((x ** 3 + 4) % o.P)
        if y is not None:
            return (x, y)


def _random_curve_point_g2(rng):
    while True:
        x = (rng.randrange(o.P), rng.randrange(o.P))
        y = o.f2sqrt(o.f2add(o.f2mul(o.f2sqr(x), x), o.B2))
        if y is not None:
            return (x, y)


def test_deserialisation(emu):
    """bls_sig_io.nim:42-99 semantics: uncompress + subgroup checks, against the oracle's definitions
    (decompression by sqrt, membership by [r]P == infinity)."""
    rng = random.Random(2024)
    inf = ctypes.c_int()
    for _ in range(4):
        sk = rng.randrange(1, o.R)
        p = o.g1_mul(o.G1_GEN, sk)
        for pt in (p, o.g1_neg(p)):
            out = buf(96)
            assert emu.emu_g1_uncompress(o.g1_compress(pt), out, ctypes.byref(inf)) == 1 and inf.value == 0
            assert out.raw == o.g1_to_blst_affine(pt)
            assert emu.emu_g1_in_subgroup(out.raw) == 1
        q = o.g2_mul(o.G2_GEN, sk)
        for pt in (q, o.g2_neg(q)):
            out = buf(192)
            assert emu.emu_g2_uncompress(o.g2_compress(pt), out, ctypes.byref(inf)) == 1 and inf.value == 0
            assert out.raw == o.g2_to_blst_affine(pt)
            assert emu.emu_g2_in_subgroup(out.raw) == 1
    # points on the curve but outside the subgroup: decompress fine, membership false (oracle: [r]P != inf)
    for _ in range(4):
        p = _random_curve_point_g1(rng)
        assert not o.g1_in_subgroup(p)
        out = buf(96)
        assert emu.emu_g1_uncompress(o.g1_compress(p), out, ctypes.byref(inf)) == 1
        assert out.raw == o.g1_to_blst_affine(p)
        assert emu.emu_g1_in_subgroup(out.raw) == 0
        q = _random_curve_point_g2(rng)
        assert not o.g2_in_subgroup(q)
        out = buf(192)
        assert emu.emu_g2_uncompress(o.g2_compress(q), out, ctypes.byref(inf)) == 1
        assert out.raw == o.g2_to_blst_affine(q)
        assert emu.emu_g2_in_subgroup(out.raw) == 0
    # clearing the cofactor of a random curve point lands in the subgroup
    hp = o.g1_mul(_random_curve_point_g1(rng), o.H1)
    assert emu.emu_g1_in_subgroup(o.g1_to_blst_affine(hp)) == 1
    # encodings
    out = buf(192)
    assert emu.emu_g2_uncompress(bytes([0xc0]) + bytes(95), out, ctypes.byref(inf)) == 1 and inf.value == 1   # serialization.nim:19-29
    bad = bytes([217, 149, 255, 97, 73, 133, 236, 43, 248, 34, 30, 10, 15, 45, 82, 72, 243, 179, 53, 17, 27, 17, 248, 180, 7, 92, 200, 153, 11, 3, 111, 137, 124, 171, 29, 218, 191, 246, 148, 57, 160, 50, 232, 129, 81, 90, 72, 161, 110, 138, 243, 116, 0, 88, 125, 180, 67, 153, 194, 181, 117, 152, 166, 147, 13, 77, 15, 91, 33, 50, 140, 199, 150, 10, 15, 10, 209, 165, 38, 57, 56, 114, 175, 29, 49, 11, 11, 126, 55, 189, 170, 46, 218, 240, 189, 144])
    assert emu.emu_g2_uncompress(bad, out, ctypes.byref(inf)) == 0                                            # serialization.nim:39-45
    good = o.g1_compress(o.g1_mul(o.G1_GEN, 5))
    out = buf(96)
    assert emu.emu_g1_uncompress(bytes([good[0] & 0x7f]) + good[1:], out, ctypes.byref(inf)) == 0           # compression flag missing
    assert emu.emu_g1_uncompress(bytes([0xc0]) + bytes(46) + b"\x01", out, ctypes.byref(inf)) == 0           # infinity with payload
    assert emu.emu_g1_uncompress(bytes([0xe0]) + bytes(47), out, ctypes.byref(inf)) == 0                     # infinity with sign bit
    xp = (o.P | (1 << 383)).to_bytes(48, "big")
    assert emu.emu_g1_uncompress(xp, out, ctypes.byref(inf)) == 0                                             # x = p
    # an x with no point on the curve
    x = 1
    while o.fp_sqrt((x ** 3 + 4) % o.P) is not None:
        x += 1
    assert emu.emu_g1_uncompress((x | (1 << 383)).to_bytes(48, "big"), out, ctypes.byref(inf)) == 0
    # tuple statuses
    pk = o.g1_compress(o.g1_mul(o.G1_GEN, 7))
    sg = o.g2_compress(o.g2_mul(o.G2_GEN, 9))
    assert emu.emu_deserialize_tuple(pk, sg) == 0
    assert emu.emu_deserialize_tuple(bytes([0xc0]) + bytes(47), sg) == 3
    assert emu.emu_deserialize_tuple(o.g1_compress(_random_curve_point_g1(rng)), sg) == 2
    assert emu.emu_deserialize_tuple(pk, o.g2_compress(_random_curve_point_g2(rng))) == 5
    assert emu.emu_deserialize_tuple(pk, bad) == 4
    assert emu.emu_deserialize_tuple(pk, bytes([0xc0]) + bytes(95)) == 0                                      # infinity signature allowed


def test_mad_census_pins_the_bench_model(emu):
    """bench.py's roofline.int_mad multiplies MAD_PER_TUPLE by the tuple rate; the table is the census of the formulas the
    kernels run (one tuple through each stage of the one-lane-per-tuple pipeline), measured here from the real code."""
    import bench
    emu.emu_mad_census.restype = ctypes.c_ulonglong
    emu.emu_mad_census.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64]
    c = [x for x in golden("batch")["cases"] if x["name"] == "n17"][0]
    rec = bytes.fromhex(c["sets"])
    names = ["k_hash_map", "k_hash_clear", "k_pkmul", "k_sig_bucket", "k_lines", "k_lineprod"]
    for i in (0, 5, 16):
        for st, name in enumerate(names):
            got = emu.emu_mad_census(st, rec[320 * i:320 * i + 320], 0x9e3779b97f4a7c15 ^ (i << 7))
            assert abs(got - bench.MAD_PER_TUPLE[name]) <= 0.02 * bench.MAD_PER_TUPLE[name], (name, got)
    assert 3.9e6 < sum(bench.MAD_PER_TUPLE.values()) < 4.1e6


def test_uncompressed_deserialisation(emu):
    """g1_deserialize / g2_deserialize (blst_pN_deserialize semantics) on the CPU build of the device code against the
    big-int oracle: round trips, compressed-in-first-half, infinity, bad encodings."""
    rng = random.Random(8)
    for _ in range(3):
        p = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
        q = o.g2_mul(o.G2_GEN, rng.randrange(1, o.R))
        inf = ctypes.c_int()
        out = buf(96)
        assert emu.emu_g1_deserialize(o.g1_serialize(p), out, ctypes.byref(inf)) == 1 and inf.value == 0
        assert out.raw == o.g1_to_blst_affine(p)
        assert emu.emu_g1_deserialize(o.g1_compress(p) + bytes(48), out, ctypes.byref(inf)) == 1 and out.raw == o.g1_to_blst_affine(p)
        out2 = buf(192)
        assert emu.emu_g2_deserialize(o.g2_serialize(q), out2, ctypes.byref(inf)) == 1 and inf.value == 0
        assert out2.raw == o.g2_to_blst_affine(q)
        assert emu.emu_g2_deserialize(o.g2_compress(q) + bytes(96), out2, ctypes.byref(inf)) == 1 and out2.raw == o.g2_to_blst_affine(q)
        bad = bytearray(o.g1_serialize(p)); bad[60] ^= 1
        assert emu.emu_g1_deserialize(bytes(bad), out, ctypes.byref(inf)) == 0
        bad = bytearray(o.g2_serialize(q)); bad[150] ^= 1
        assert emu.emu_g2_deserialize(bytes(bad), out2, ctypes.byref(inf)) == 0
        assert emu.emu_deserialize_tuple_ex(o.g1_serialize(p), o.g2_serialize(q), 3) == 0
        assert emu.emu_deserialize_tuple_ex(o.g1_serialize(p), o.g2_compress(q), 1) == 0
    inf = ctypes.c_int()
    out, out2 = buf(96), buf(192)
    assert emu.emu_g1_deserialize(bytes([0x40]) + bytes(95), out, ctypes.byref(inf)) == 1 and inf.value == 1
    assert emu.emu_g1_deserialize(bytes([0x40]) + bytes(94) + b"\x01", out, ctypes.byref(inf)) == 0
    assert emu.emu_g1_deserialize(bytes(96), out, ctypes.byref(inf)) == 0
    assert emu.emu_g2_deserialize(bytes([0x40]) + bytes(191), out2, ctypes.byref(inf)) == 1 and inf.value == 1
    assert emu.emu_g2_deserialize(bytes([0x60]) + bytes(191), out2, ctypes.byref(inf)) == 0
    assert emu.emu_g1_deserialize(o.P.to_bytes(48, "big") + bytes(48), out, ctypes.byref(inf)) == 0


def test_lane_cooperative_fp12_engine(emu):
    """k_tail's Fp12 product / square (c12.hpp: 108 or 63 one-multiplication items, 168 limb-combination items, 12 reductions)
    executed item by item on the CPU, bounds tracked, against the tower's fp12_mul and the big-int oracle; a chain of
    squarings and products keeps the bounds."""
    from util import fp12_to_bytes
    rng = random.Random(12)

    def rnd12():
        return tuple((rng.randrange(o.P), rng.randrange(o.P)) for _ in range(6))
    for _ in range(3):
        a, b = rnd12(), rnd12()
        A, B = fp12_to_bytes(a), fp12_to_bytes(b)
        got = call(emu, "emu_c12_mul", A, B, outlen=576)
        assert got == call(emu, "emu_fp12_mul", A, B, outlen=576)
        assert fp12_from_bytes(got) == o.f12mul(a, b)
        assert fp12_from_bytes(call(emu, "emu_c12_sqr", A, outlen=576)) == o.f12sqr(a)
        # the engine called with explicit product / square flag (limbs stay in their lanes: carries by neighbour, quotient from lane 13)
        assert fp12_from_bytes(call(emu, "emu_c12_rowphase", A, B, 0, outlen=576)) == o.f12mul(a, b)
        assert fp12_from_bytes(call(emu, "emu_c12_rowphase", A, A, 1, outlen=576)) == o.f12sqr(a)
    # extreme operands: p - 1 in every coefficient (largest canonical limbs and values)
    top = fp12_to_bytes(tuple((o.P - 1, o.P - 1) for _ in range(6)))
    x = fp12_to_bytes(rnd12())
    want = fp12_from_bytes(x)
    for i in range(8):                                   # a chain on the row phase: semi-normalised limbs feed the next product
        x = call(emu, "emu_c12_rowphase", x, x, i & 1, outlen=576)
        want = o.f12sqr(want)
    assert fp12_from_bytes(x) == want
    assert fp12_from_bytes(call(emu, "emu_c12_rowphase", top, top, 1, outlen=576)) == o.f12sqr(fp12_from_bytes(top))
    x = fp12_to_bytes(rnd12())
    want = fp12_from_bytes(x)
    for i in range(6):
        x = call(emu, "emu_c12_sqr", x, outlen=576)
        want = o.f12sqr(want)
        y = call(emu, "emu_c12_mul", x, x, outlen=576)
        assert fp12_from_bytes(y) == o.f12sqr(want)
    assert fp12_from_bytes(x) == want


def test_team_formulas_equal_the_plain_ones(emu):
    """jac_dbl_team / miller_dbl_step_team (the lane-cooperative kernels' formulas, curve.hpp / pairing.hpp) with the solo team,
    bounds tracked: same points as jac_dbl, same 68 lines as miller_lines."""
    rng = random.Random(21)
    for _ in range(3):
        p = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
        q = o.g2_mul(o.G2_GEN, rng.randrange(1, o.R))
        P, Q = g1_aff_to_jac_bytes(p), g2_aff_to_jac_bytes(q)
        for _ in range(3):                                   # a few doublings deep: non-trivial Z
            a, b = call(emu, "emu_g2_dbl_team", Q, outlen=288), call(emu, "emu_g2_dbl", Q, outlen=288)
            assert g2_jac_to_affine(a) == g2_jac_to_affine(b) == o.g2_add(g2_jac_to_affine(Q), g2_jac_to_affine(Q))
            Q = a
            a, b = call(emu, "emu_g1_dbl_team", P, outlen=144), call(emu, "emu_g1_dbl", P, outlen=144)
            assert g1_jac_to_affine(a) == g1_jac_to_affine(b)
            P = a
        # the team addition: general position (non-trivial Z on both sides), P + P, P + (-P), infinity on either side
        q2 = o.g2_mul(o.G2_GEN, rng.randrange(1, o.R))
        Q2 = call(emu, "emu_g2_dbl", g2_aff_to_jac_bytes(q2), outlen=288)
        for A, B in ((Q, Q2), (Q2, Q), (Q, Q), (Q, g2_aff_to_jac_bytes(o.g2_neg(g2_jac_to_affine(Q)))), (bytes(288), Q), (Q, bytes(288))):
            a, b = call(emu, "emu_g2_add_team", A, B, outlen=288), call(emu, "emu_g2_add", A, B, outlen=288)
            assert g2_jac_to_affine(a) == g2_jac_to_affine(b)
        # the same in G1 (the lane-team segment reduction of the Pippenger path)
        p2 = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
        P2 = call(emu, "emu_g1_dbl", g1_aff_to_jac_bytes(p2), outlen=144)
        for A, B in ((P, P2), (P2, P), (P, P), (P, g1_aff_to_jac_bytes(o.g1_neg(g1_jac_to_affine(P)))), (bytes(144), P), (P, bytes(144)), (bytes(144), bytes(144))):
            a, b = call(emu, "emu_g1_add_team", A, B, outlen=144), call(emu, "emu_g1_add", A, B, outlen=144)
            assert g1_jac_to_affine(a) == g1_jac_to_affine(b)
