"""Pins the C restatement (oracle/bls_oracle.c) to the golden fixtures, which carry the reference's
KATs through the Python oracle: hash-to-G2 bytes, sk->pk, signatures, per-stage batch values,
verdicts of every tests/t_batch_verifier.nim scenario, G1 sums, MSM, fastAggregateVerify."""
import bls12381_py as o
import c_oracle as co
from util import fp12_from_bytes, fp12_hexlist_to_flat, golden


def test_hash_to_g2_bytes():
    for v in golden("h2c"):
        assert co.hash_to_g2(bytes.fromhex(v["msg"]), v["dst"].encode()).hex() == v["h"]


def test_sk_to_pk_and_sign_kats():
    # tests/priv_to_pub.nim:32-35 ; tests/eth2_vectors.nim:33-47 (PoP = sign under DST_POP is pinned in python;
    # here the C signer under DST_SIG is cross-checked against the python oracle)
    assert o.g1_compress(o.g1_from_blst_affine(co.sk_to_pk(1000))).hex() == \
        "a60e75190e62b6a54142d147289a735c4ce11a9d997543da539a3db57def5ed83ba40b74e55065f02b35aa1d504c404b"
    sk = int("263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3", 16)
    msg = o.sha256(b"message")
    assert o.g2_from_blst_affine(co.sign(sk, msg)) == o.sign(sk, msg)


def test_batch_fixtures():
    for c in golden("batch")["cases"]:
        rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
        for mode, nt in (("serial", 0), ("chunks4", 4)):
            if mode not in c:
                continue
            st = c[mode]
            if "gt" in st:
                ok, got = co.batch_verify(rec, rnd, nt, stages=True)
                assert [str(x) for x in got["r"]] == st["r"]
                n = c["n"]
                assert [got["H"][192 * i:192 * i + 192].hex() for i in range(n)] == st["H"]
                assert [got["rPK"][96 * i:96 * i + 96].hex() for i in range(n)] == st["rPK"]
                assert got["aggsig"].hex() == st["aggsig"]
                assert fp12_from_bytes(got["gt"]) == fp12_hexlist_to_flat(st["gt"])
            else:
                ok = co.batch_verify(rec, rnd, nt)
            assert ok == c["expect"], (c["name"], mode)
    assert co.batch_verify(b"", bytes(32), 4) is False


def test_make_batch_is_valid_and_tamper_detected():
    rec = co.make_batch(12, seed=5)
    rnd = o.sha256(b"Mr F was here")
    assert co.batch_verify(rec, rnd, 4) and co.batch_verify(rec, rnd, 0)
    bad = bytearray(rec)
    bad[320 * 7 + 100] ^= 2
    assert not co.batch_verify(bytes(bad), rnd, 4)
    # python oracle agrees on a prefix
    sets = [(o.g1_from_blst_affine(rec[320 * i:320 * i + 96]), rec[320 * i + 96:320 * i + 128],
             o.g2_from_blst_affine(rec[320 * i + 128:320 * i + 320])) for i in range(3)]
    assert o.batch_verify(sets, rnd, 4)


def test_g1_sum_msm_fav():
    g = golden("msm")
    a = g["aggregate"]
    assert co.g1_sum(bytes.fromhex(a["points"])).hex() == a["sum_affine"]
    for v in g["msm"]:
        if v["n"] <= 33:
            assert co.msm_g1(bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"]), v["nbits"]).hex() == v["result_affine"]
    for v in golden("fav"):
        pks, msg = bytes.fromhex(v["pks"]), bytes.fromhex(v["msg"])
        assert co.fast_aggregate_verify(pks, msg, bytes.fromhex(v["sig"])) is True
        assert co.fast_aggregate_verify(pks, msg, bytes.fromhex(v["bad_sig"])) is False
    assert co.fast_aggregate_verify(b"", b"x", bytes(192)) is False


def test_compress_deserialize_roundtrip_and_python_parity():
    rec = co.make_batch(6, seed=3)
    pk, ms, sg = co.compress_sets(rec)
    for i in range(6):
        assert pk[48 * i:48 * i + 48] == o.g1_compress(o.g1_from_blst_affine(rec[320 * i:320 * i + 96]))
        assert sg[96 * i:96 * i + 96] == o.g2_compress(o.g2_from_blst_affine(rec[320 * i + 128:320 * i + 320]))
    ok, out, st = co.deserialize_sets(pk, ms, sg)
    assert ok and out == rec and st == bytes(6)
    # reference KAT: malformed signature must be rejected (tests/serialization.nim:39-45)
    bad = bytes([217, 149, 255, 97, 73, 133, 236, 43, 248, 34, 30, 10, 15, 45, 82, 72, 243, 179, 53, 17, 27, 17, 248, 180, 7, 92, 200, 153, 11, 3, 111, 137, 124, 171, 29, 218, 191, 246, 148, 57, 160, 50, 232, 129, 81, 90, 72, 161, 110, 138, 243, 116, 0, 88, 125, 180, 67, 153, 194, 181, 117, 152, 166, 147, 13, 77, 15, 91, 33, 50, 140, 199, 150, 10, 15, 10, 209, 165, 38, 57, 56, 114, 175, 29, 49, 11, 11, 126, 55, 189, 170, 46, 218, 240, 189, 144])
    ok, out, st = co.deserialize_sets(pk[:48], ms[:32], bad)
    assert not ok and st == bytes([4])
    ok, out, st = co.deserialize_sets(bytes([0xc0]) + bytes(47), ms[:32], sg[:96])
    assert not ok and st == bytes([3])
    ok, out, st = co.deserialize_sets(pk[:48], ms[:32], bytes([0xc0]) + bytes(95))
    assert ok and st == bytes([0]) and out[128:] == bytes(192)


def test_pippenger_restatement_equals_naive_msm():
    """oracle_msm_g1_pippenger (CPU baseline of the MSM bench row) == the naive per-point multiplication == golden."""
    import random
    for v in golden("msm")["msm"]:
        pts, sc = bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"])
        assert co.msm_g1_pippenger(pts, sc, v["nbits"]) == co.msm_g1(pts, sc, v["nbits"])
        assert co.msm_g1_pippenger(pts, sc, v["nbits"]).hex() == v["result_affine"]
    v = golden("msm")["msm"][-1]
    pts, sc = bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"])
    rng = random.Random(3)
    big = b"".join(pts[96 * rng.randrange(v["n"]):][:96] for _ in range(3000))
    bsc = bytes(rng.getrandbits(8) for _ in range(32 * 3000))
    for nbits in (255, 64, 9):
        assert co.msm_g1_pippenger(big, bsc, nbits) == co.msm_g1(big, bsc, nbits)


def test_core_verify():
    sk = 0x1234567
    pk, msg = co.sk_to_pk(sk), b"Mr F was here"
    sig = co.sign(sk, msg)
    assert co.core_verify(pk, msg, sig) and not co.core_verify(pk, msg + b"!", sig)


def test_combine_and_g2_msm_restatements():
    """oracle_combine against the fixture made by the KAT-pinned python oracle (scalars in word order 3,2,1,0, combined key and
    signature), and oracle_msm_g2 against python big-int G2 arithmetic."""
    import random
    g = golden("batch")["combine"]
    pk, sg, sc = co.combine(bytes.fromhex(g["rnd"]), bytes.fromhex(g["pks"]), bytes.fromhex(g["sigs"]))
    assert [str(x) for x in sc] == g["scalars"] and pk.hex() == g["out_pk"] and sg.hex() == g["out_sig"]
    rng = random.Random(12)
    sigs = bytes.fromhex(g["sigs"])[:192 * 5]
    Q = [o.g2_from_blst_affine(sigs[192 * i:192 * i + 192]) for i in range(5)]
    for nbits, sb in ((64, 8), (255, 32), (13, 2)):
        K = [rng.getrandbits(8 * sb) for _ in range(5)]
        want = None
        for q, k in zip(Q, K):
            want = o.g2_add(want, o.g2_mul(q, k % (1 << nbits)))
        got = co.msm_g2(sigs, b"".join(k.to_bytes(sb, "little") for k in K), nbits, sb)
        assert got == o.g2_to_blst_affine(want)


def test_g2_pippenger_restatement_equals_naive():
    """oracle_msm_g2_pippenger (checker of the device G2 MSM at 2^18 points) == the naive loop, which the test above pins to the
    python oracle; both scalar spacings."""
    import random
    g = golden("batch")["combine"]
    sigs = bytes.fromhex(g["sigs"])
    k = len(sigs) // 192
    rng = random.Random(21)
    n = 700
    pts = b"".join(sigs[192 * rng.randrange(k):][:192] for _ in range(n))
    for nbits, sb in ((64, 8), (255, 32), (130, 17)):
        sc = bytes(rng.getrandbits(8) for _ in range(sb * n))
        assert co.msm_g2_pippenger(pts, sc, nbits, sb) == co.msm_g2(pts, sc, nbits, sb), nbits


def test_aggregate_verify_restatement():
    """oracle_aggregate_verify (checker of the device aggregateVerify at 1 024 ... 20 000 pairs) against the KAT-pinned python oracle:
    verdicts on the n9 fixture (valid, rotated messages, missing pair, infinity key), the forged pair (passes: no blinding), messages
    of other lengths, and the GT value == the python oracle's final exponentiation of the same product."""
    c = [x for x in golden("batch")["cases"] if x["name"] == "n9"][0]
    rec = bytes.fromhex(c["sets"])
    n = c["n"]
    pks = [rec[320 * i:320 * i + 96] for i in range(n)]
    msgs = [rec[320 * i + 96:320 * i + 128] for i in range(n)]
    sigs = b"".join(rec[320 * i + 128:320 * i + 320] for i in range(n))
    agg = co.g2_sum(sigs)
    assert agg == o.g2_to_blst_affine(o.aggregate_g2([o.g2_from_blst_affine(sigs[192 * i:192 * i + 192]) for i in range(n)]))
    ok, gt = co.aggregate_verify(pks, msgs, agg, gt=True)
    assert ok is True
    assert o.aggregate_verify([o.g1_from_blst_affine(p) for p in pks], msgs, o.g2_from_blst_affine(agg)) is True
    assert co.aggregate_verify(pks, msgs[1:] + msgs[:1], agg) is False
    assert co.aggregate_verify(pks[:-1], msgs[:-1], agg) is False
    assert co.aggregate_verify([bytes(96)] + pks[1:], msgs, agg) is False
    f = [x for x in golden("batch")["cases"] if x["name"] == "forged_pair"][0]
    fr = bytes.fromhex(f["sets"])
    fp, fm = [fr[320 * i:320 * i + 96] for i in range(f["n"])], [fr[320 * i + 96:320 * i + 128] for i in range(f["n"])]
    assert co.aggregate_verify(fp, fm, co.g2_sum(b"".join(fr[320 * i + 128:320 * i + 320] for i in range(f["n"])))) is True
    texts = [b"", b"a", b"Mr F was here", bytes(200)]
    keys = [o.keygen_seed(i) for i in range(4)]
    sig = o.g2_to_blst_affine(o.aggregate_g2([o.sign(sk, t) for (pk, sk), t in zip(keys, texts)]))
    pkb = [o.g1_to_blst_affine(pk) for pk, sk in keys]
    assert co.aggregate_verify(pkb, texts, sig) is True and co.aggregate_verify(pkb, [b"x"] + texts[1:], sig) is False
