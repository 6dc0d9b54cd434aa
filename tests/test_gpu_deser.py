"""GPU parity: batched deserialisation + validation (bls_sig_io.nim:42-99) and the wire-format batch verify."""
import random

import pytest

import bls12381_py as o
from util import golden

pytestmark = pytest.mark.gpu

BAD_SIG = bytes([217, 149, 255, 97, 73, 133, 236, 43, 248, 34, 30, 10, 15, 45, 82, 72, 243, 179, 53, 17, 27, 17, 248, 180, 7, 92, 200, 153, 11, 3, 111, 137, 124, 171, 29, 218, 191, 246, 148, 57, 160, 50, 232, 129, 81, 90, 72, 161, 110, 138, 243, 116, 0, 88, 125, 180, 67, 153, 194, 181, 117, 152, 166, 147, 13, 77, 15, 91, 33, 50, 140, 199, 150, 10, 15, 10, 209, 165, 38, 57, 56, 114, 175, 29, 49, 11, 11, 126, 55, 189, 170, 46, 218, 240, 189, 144])


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="module")
def cache(m):
    return m.BatchedBLSVerifierCache.init(max_sets=4096, numThreads=4)


def _curve_point_g1(rng):
    while True:
        x = rng.randrange(o.P)
        y = o.fp_sqrt((x ** 3 + 4) % o.P)
        if y is not None:
            return (x, y)


def _curve_point_g2(rng):
    while True:
        x = (rng.randrange(o.P), rng.randrange(o.P))
        y = o.f2sqrt(o.f2add(o.f2mul(o.f2sqr(x), x), o.B2))
        if y is not None:
            return (x, y)


def test_golden_cases_through_wire_format(m, cache):
    """every tests/t_batch_verifier.nim scenario, fed as compressed bytes"""
    for c in golden("batch")["cases"]:
        rec, rnd, n = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"]), c["n"]
        pts = [(o.g1_from_blst_affine(rec[320 * i:320 * i + 96]), rec[320 * i + 96:320 * i + 128],
                o.g2_from_blst_affine(rec[320 * i + 128:320 * i + 320])) for i in range(n)]
        pk = b"".join(o.g1_compress(p) for p, _, _ in pts)
        ms = b"".join(mm for _, mm, _ in pts)
        sg = b"".join(o.g2_compress(s) for _, _, s in pts)
        ok, out, st = m.deserializeSets(cache, pk, ms, sg)
        if c["name"] == "inf_pk":
            assert not ok and st[1] == 3
            v, st2 = m.batchVerifyCompressed(cache, pk, ms, sg, rnd)
            assert v is False and st2 == st
            continue
        assert ok and st == bytes(n) and out == rec          # byte-exact SignatureSet records
        v, st2 = m.batchVerifyCompressed(cache, pk, ms, sg, rnd)
        assert v == c["expect"] and st2 == bytes(n)


def test_invalid_encodings_and_subgroup(m, cache):
    rng = random.Random(77)
    good_pk = o.g1_compress(o.g1_mul(o.G1_GEN, 1234567))
    good_sg = o.g2_compress(o.g2_mul(o.G2_GEN, 7654321))
    msg = bytes(range(32))
    x = 1
    while o.fp_sqrt((x ** 3 + 4) % o.P) is not None:
        x += 1
    cases = [
        (good_pk, good_sg, 0),
        (bytes([good_pk[0] & 0x7f]) + good_pk[1:], good_sg, 1),                 # not flagged compressed
        ((o.P | (1 << 383)).to_bytes(48, "big"), good_sg, 1),                   # x = p
        ((x | (1 << 383)).to_bytes(48, "big"), good_sg, 1),                     # x not on the curve
        (bytes([0xc0]) + bytes(46) + b"\x01", good_sg, 1),                      # infinity with payload
        (bytes([0xc0]) + bytes(47), good_sg, 3),                                # infinity public key
        (o.g1_compress(_curve_point_g1(rng)), good_sg, 2),                      # on the curve, outside G1
        (good_pk, BAD_SIG, 4),                                                  # tests/serialization.nim:39-45
        (good_pk, bytes([good_sg[0] & 0x7f]) + good_sg[1:], 4),
        (good_pk, o.g2_compress(_curve_point_g2(rng)), 5),                      # on the curve, outside G2
        (good_pk, bytes([0xc0]) + bytes(95), 0),                                # infinity signature is allowed
        (o.g1_compress(o.g1_neg(o.g1_mul(o.G1_GEN, 99))), o.g2_compress(o.g2_neg(o.g2_mul(o.G2_GEN, 5))), 0),
    ]
    pk = b"".join(c[0] for c in cases)
    sg = b"".join(c[1] for c in cases)
    ok, out, st = m.deserializeSets(cache, pk, msg * len(cases), sg)
    assert list(st) == [c[2] for c in cases] and not ok
    for i, c in enumerate(cases):
        r = out[320 * i:320 * i + 320]
        if c[2] == 0:
            assert r[:96] == o.g1_to_blst_affine(o.g1_decompress(c[0])) and r[128:] == o.g2_to_blst_affine(o.g2_decompress(c[1]))
        else:
            assert r[:96] == bytes(96) and r[128:] == bytes(192)
    assert m.deserializeSets(cache, b"", b"", b"") == (True, b"", b"")
    assert m.batchVerifyCompressed(cache, b"", b"", b"", bytes(32)) == (False, b"")


def test_scale_vs_c_oracle(m, cache):
    """4096 distinct valid tuples + sprinkled invalid ones: statuses and records equal the C restatement's
    (decompression by square root, membership by [r]P)."""
    import c_oracle as co
    rng = random.Random(5)
    n = 4096
    rec = co.make_batch(n, seed=11)
    pk, ms, sg = co.compress_sets(rec)
    rnd = o.sha256(b"Mr F was here")
    v, st = m.batchVerifyCompressed(cache, pk, ms, sg, rnd)
    assert v is True and st == bytes(n)
    print("deser ms:", m.lib().mi355_bls_last_deser_ms(cache._h), cache.timings())
    pkb, sgb = bytearray(pk), bytearray(sg)
    bad_pk = o.g1_compress(_curve_point_g1(rng))
    bad_sg = o.g2_compress(_curve_point_g2(rng))
    for i in (3, 1000, 4095):
        pkb[48 * i:48 * i + 48] = bad_pk
    for i in (7, 2048):
        sgb[96 * i:96 * i + 96] = bad_sg
    sgb[96 * 100:96 * 100 + 96] = BAD_SIG
    okc, outc, stc = co.deserialize_sets(bytes(pkb), ms, bytes(sgb))
    ok, out, st = m.deserializeSets(cache, bytes(pkb), ms, bytes(sgb))
    assert (ok, st) == (okc, stc) and not ok
    for i in range(n):
        if st[i] == 0:
            assert out[320 * i:320 * i + 320] == outc[320 * i:320 * i + 320]
    v, st2 = m.batchVerifyCompressed(cache, bytes(pkb), ms, bytes(sgb), rnd)
    assert v is False and st2 == st
    # a valid encoding of a different (valid) signature: deserialises, batch must not verify
    sgc = bytearray(sg)
    sgc[0:96], sgc[96:192] = sg[96:192], sg[0:96]
    v, st3 = m.batchVerifyCompressed(cache, pk, ms, bytes(sgc), rnd)
    assert v is False and st3 == bytes(n)


# ---- the other forms of fromBytes: 96- / 192-byte images (blst_pN_deserialize) and fromBytesKnownOnCurve ----

def test_serialization_roundtrip_all_forms(m, cache):
    """The shape of tests/serialization.nim:52-142: keys of the secret keys 1000..1004, signatures over the hash-to-curve
    spec messages; serialize -> fromBytes gives the same point, for the compressed AND the uncompressed form of each side
    (the reference's uncompressed half is commented out, :97-109; its entry points exist, bls_sig_io.nim:49-52,88-91)."""
    import c_oracle as co
    msgs = [b"", b"abc", b"abcdef0123456789", b"q128_" + b"q" * 128, b"a512_" + b"a" * 512]
    sets, m32 = [], []
    for sk in range(1000, 1005):
        pk = co.sk_to_pk(sk)
        for msg in msgs:
            d = o.sha256(msg)                                   # SignatureSet messages are 32-byte digests (bls_batch_verifier.nim:42)
            sets.append(pk + d + co.sign(sk, d))
            m32.append(d)
    rec = b"".join(sets)
    n = len(sets)
    pk48, ms, sg96 = co.compress_sets(rec)
    pk96, sg192 = co.serialize_sets(rec)
    for i in (0, 7):                                            # the C serialiser against the KAT-pinned python one
        assert pk96[96 * i:96 * i + 96] == o.g1_serialize(o.g1_from_blst_affine(rec[320 * i:320 * i + 96]))
        assert sg192[192 * i:192 * i + 192] == o.g2_serialize(o.g2_from_blst_affine(rec[320 * i + 128:320 * i + 320]))
    for pku, sgu in ((False, False), (True, False), (False, True), (True, True)):
        for known in (False, True):
            ok, out, st = m.deserializeSetsEx(cache, pk96 if pku else pk48, ms, sg192 if sgu else sg96, pk_uncompressed=pku, sig_uncompressed=sgu,
                                              known_on_curve=known)
            assert ok and st == bytes(n) and out == rec, (pku, sgu, known)
    # blst_pN_deserialize also takes a COMPRESSED encoding in its first half (top bit set)
    pk_mixed = b"".join(pk48[48 * i:48 * i + 48] + bytes(48) if i % 2 else pk96[96 * i:96 * i + 96] for i in range(n))
    sg_mixed = b"".join(sg96[96 * i:96 * i + 96] + bytes(96) if i % 3 else sg192[192 * i:192 * i + 192] for i in range(n))
    ok, out, st = m.deserializeSetsEx(cache, pk_mixed, ms, sg_mixed, pk_uncompressed=True, sig_uncompressed=True)
    assert ok and out == rec


def test_uncompressed_rejections_match_the_oracles(m, cache):
    """Per-tuple statuses of the 96- / 192-byte forms against the C restatement and the big-int oracle: off-curve points,
    coordinates >= p, flag-bit misuse, infinity encodings (good and bad), points outside the subgroups (rejected by
    fromBytes, accepted by fromBytesKnownOnCurve), the reference's BAD_SIG bytes (tests/serialization.nim:39-45)."""
    import c_oracle as co
    rng = random.Random(77)
    base = co.make_batch(4, seed=9)
    gpk96, gsg192 = co.serialize_sets(base)
    good_pk, good_sg = gpk96[:96], gsg192[:192]
    p_off = _curve_point_g1(rng)                                 # on the curve, (almost surely) not in G1
    q_off = _curve_point_g2(rng)
    P = o.P
    cases = []                                                   # (pk96, sig192)
    cases.append((good_pk, good_sg))
    cases.append((o.g1_serialize(p_off), good_sg))               # pk not in G1
    cases.append((good_pk, o.g2_serialize(q_off)))               # sig not in G2
    bad_y = bytearray(good_pk); bad_y[95] ^= 1
    cases.append((bytes(bad_y), good_sg))                        # off the curve
    cases.append((P.to_bytes(48, "big") + good_pk[48:], good_sg))              # x = p: top bits 000 but >= p
    cases.append((good_pk[:48] + (P + 1).to_bytes(48, "big"), good_sg))        # y >= p
    cases.append((bytes([0x40]) + bytes(95), good_sg))           # infinity public key: decodes, then rejected (status 3)
    cases.append((bytes([0x40]) + bytes(94) + b"\x01", good_sg))                # bad infinity
    cases.append((bytes([0x20 | good_pk[0]]) + good_pk[1:], good_sg))          # sign flag without the compressed bit
    cases.append((good_pk, bytes([0x40]) + bytes(191)))          # infinity signature: allowed
    cases.append((good_pk, bytes([0x40]) + bytes(100) + b"\x02" + bytes(90)))  # bad infinity signature
    bad_sy = bytearray(good_sg); bad_sy[191] ^= 4
    cases.append((good_pk, bytes(bad_sy)))                       # signature off the curve
    cases.append((good_pk, good_sg[:96] + P.to_bytes(48, "big") + good_sg[144:]))   # y.c1 = p
    cases.append((good_pk, BAD_SIG + bytes(96)))                 # compressed-form bytes of the reference's hardening test
    cases.append((bytes(96), good_sg))                           # all-zero: (0, 0) is not on the curve
    pk = b"".join(c[0] for c in cases)
    sg = b"".join(c[1] for c in cases)
    ms = bytes(32 * len(cases))
    for known in (False, True):
        flags = 3 | (4 if known else 0)
        ok_c, out_c, st_c = co.deserialize_sets_ex(pk, ms, sg, flags)
        ok, out, st = m.deserializeSetsEx(cache, pk, ms, sg, pk_uncompressed=True, sig_uncompressed=True, known_on_curve=known)
        assert (ok, st) == (ok_c, st_c), (known, list(st), list(st_c))
        assert out == out_c
        want = [0, 2, 5, 1, 1, 1, 3, 1, 1, 0, 4, 4, 4, 4, 1] if not known else [0, 0, 0, 1, 1, 1, 3, 1, 1, 0, 4, 4, 4, 4, 1]
        assert list(st) == want
    # the big-int oracle on the decode step alone
    for pkb, sgb in cases:
        try:
            o.g1_deserialize(pkb); pk_ok = True
        except ValueError:
            pk_ok = False
        try:
            o.g2_deserialize(sgb); sg_ok = True
        except ValueError:
            sg_ok = False
        _, _, st1 = m.deserializeSetsEx(cache, pkb, bytes(32), sgb, pk_uncompressed=True, sig_uncompressed=True, known_on_curve=True)
        assert (st1[0] == 1) == (not pk_ok)
        if pk_ok and st1[0] != 3:
            assert (st1[0] == 4) == (not sg_ok)


def test_known_on_curve_compressed_form(m, cache):
    """fromBytesKnownOnCurve on the 48- / 96-byte forms (bls_sig_io.nim:60-79, 101-121): a curve point outside the subgroup
    passes, the infinity public key and bad encodings still fail."""
    import c_oracle as co
    rng = random.Random(5)
    base = co.make_batch(2, seed=3)
    pk48, ms, sg96 = co.compress_sets(base)
    p_off, q_off = _curve_point_g1(rng), _curve_point_g2(rng)
    pk = pk48[:48] + o.g1_compress(p_off) + pk48[48:96] + bytes([0xc0]) + bytes(47)
    sg = sg96[:96] + sg96[96:192] + o.g2_compress(q_off) + sg96[:96]
    ms4 = ms + ms
    ok, out, st = m.deserializeSetsEx(cache, pk, ms4, sg, known_on_curve=True)
    assert not ok and list(st) == [0, 0, 0, 3]
    assert (False, out, st) == co.deserialize_sets_ex(pk, ms4, sg, 4)
    ok, out, st = m.deserializeSetsEx(cache, pk, ms4, sg)
    assert list(st) == [0, 2, 5, 3]


def test_headline_size_distinct_tuples_vs_c_oracle(m):
    """fromBytes at the headline batch size: 65 536 DISTINCT tuples (device signer) with invalid encodings sprinkled over the
    batch - statuses and every surviving record equal the C restatement's (decompression by square root, subgroup membership by
    [r]P), and the wire-format batch verify of the clean batch is true."""
    import torch
    import bench
    import c_oracle as co
    rng = random.Random(65536)
    n = 65536
    dev = torch.device("cuda", 0)
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    rec = bytes(bench.sign_records(m, cache, dev, range(5_000_000, 5_000_000 + n)).cpu().numpy())
    pk, ms, sg = co.compress_sets(rec)
    rnd = o.sha256(b"Mr F was here")
    v, st = m.batchVerifyCompressed(cache, pk, ms, sg, rnd)
    assert v is True and st == bytes(n)
    pkb, sgb = bytearray(pk), bytearray(sg)
    bad_pk = [o.g1_compress(_curve_point_g1(rng)) for _ in range(3)]            # on the curve, not in G1
    bad_sg = [o.g2_compress(_curve_point_g2(rng)) for _ in range(3)]
    touched = set()
    for j in range(40):
        i = rng.randrange(n)
        touched.add(i)
        kind = j % 5
        if kind == 0:
            pkb[48 * i:48 * i + 48] = bad_pk[j % 3]
        elif kind == 1:
            sgb[96 * i:96 * i + 96] = bad_sg[j % 3]
        elif kind == 2:
            pkb[48 * i:48 * i + 48] = bytes([0xc0]) + bytes(47)                 # infinity key
        elif kind == 3:
            pkb[48 * i] &= 0x7f                                                 # compression bit cleared
        else:
            sgb[96 * i:96 * i + 96] = BAD_SIG
    okc, outc, stc = co.deserialize_sets(bytes(pkb), ms, bytes(sgb))
    ok, out, st = m.deserializeSets(cache, bytes(pkb), ms, bytes(sgb))
    assert (ok, st) == (okc, stc) and not ok
    assert all(st[i] != 0 for i in touched) and sum(1 for x in st if x) == len(touched)
    good = [i for i in range(n) if st[i] == 0]
    assert all(out[320 * i:320 * i + 320] == outc[320 * i:320 * i + 320] for i in good)
    assert all(out[320 * i:320 * i + 320] == rec[320 * i:320 * i + 320] for i in good[::97])
    cache.close()
