"""GPU parity: G1 aggregateAll and fastAggregateVerify (bls_sig_min_pubkey.nim:234-258) through the C ABI."""
import hashlib

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
    return m.BatchedBLSVerifierCache.init(max_sets=16384)


def test_aggregate_golden(m, cache):
    a = golden("msm")["aggregate"]
    pts = bytes.fromhex(a["points"])
    out = m.aggregateAll(cache, pts)
    assert o.g1_to_blst_affine(g1_jac_to_affine(out)).hex() == a["sum_affine"]
    for n in (1, 2, 63, 64, 65, 200):
        sub = pts[:96 * n]
        want = o.aggregate_g1([o.g1_from_blst_affine(sub[96 * i:96 * i + 96]) for i in range(n)])
        assert g1_jac_to_affine(m.aggregateAll(cache, sub)) == want
    assert m.aggregateAll(cache, b"") is None
    # P + (-P) + Q = Q ; doubling inside the reduction: P + P
    p = o.g1_from_blst_affine(pts[:96])
    q = o.g1_from_blst_affine(pts[96:192])
    buf = o.g1_to_blst_affine(p) + o.g1_to_blst_affine(o.g1_neg(p)) + o.g1_to_blst_affine(q)
    assert g1_jac_to_affine(m.aggregateAll(cache, buf)) == q
    assert g1_jac_to_affine(m.aggregateAll(cache, pts[:96] * 2)) == o.g1_add(p, p)
    assert g1_jac_to_affine(m.aggregateAll(cache, o.g1_to_blst_affine(p) + o.g1_to_blst_affine(o.g1_neg(p)))) is None


def test_fast_aggregate_verify_golden(m, cache):
    for v in golden("fav"):
        pks, msg = bytes.fromhex(v["pks"]), bytes.fromhex(v["msg"])
        assert m.fastAggregateVerify(cache, pks, msg, bytes.fromhex(v["sig"])) is True
        assert m.fastAggregateVerify(cache, pks, msg, bytes.fromhex(v["bad_sig"])) is False
        assert m.fastAggregateVerify(cache, pks, msg + b"x", bytes.fromhex(v["sig"])) is False
    assert m.fastAggregateVerify(cache, b"", b"m", bytes(192)) is False     # empty -> false


def test_fast_aggregate_verify_32768(m, cache):
    """Config 3 shape: 32 768 distinct keys, one message, aggregate signature (C restatement as generator
    and checker): same aggregate point bit-exactly, same verdicts."""
    import c_oracle as co
    n = 32768
    sks = [int.from_bytes(hashlib.sha256(b"fav" + i.to_bytes(4, "little")).digest(), "little") % o.R or 1 for i in range(n)]
    pks = b"".join(co.sk_to_pk(sk) for sk in sks)
    msg = b"Mr F was here"
    hm = co.hash_to_g2(msg, o.DST_SIG)
    sig = co.g2_mul(hm, sum(sks) % o.R)
    bad = co.g2_mul(hm, (sum(sks) + 1) % o.R)
    assert o.g1_to_blst_affine(g1_jac_to_affine(m.aggregateAll(cache, pks))) == co.g1_sum(pks)
    assert m.fastAggregateVerify(cache, pks, msg, sig) is True
    print("timings(ms):", cache.timings())
    assert m.fastAggregateVerify(cache, pks, msg, bad) is False
    assert co.fast_aggregate_verify(pks, msg, sig) is True and co.fast_aggregate_verify(pks, msg, bad) is False
    # dropping one key breaks it
    assert m.fastAggregateVerify(cache, pks[96:], msg, sig) is False
    # the same keys sharded over 1, 3 and 8 contexts (devices in a deployment): same verdicts, same GT value
    gt = cache.fetch(4, 576) if m.fastAggregateVerify(cache, pks, msg, sig) else None
    for ngpu in (1, 3, 8):
        caches = [m.BatchedBLSVerifierCache.init(max_sets=64) for _ in range(ngpu)]
        assert m.fastAggregateVerifyMulti(caches, pks, msg, sig) is True
        assert caches[0].fetch(4, 576) == gt
        assert m.fastAggregateVerifyMulti(caches, pks, msg, bad) is False
        assert m.fastAggregateVerifyMulti(caches, pks[96:], msg, sig) is False
        assert m.fastAggregateVerifyMulti(caches, pks[:96 * 2], msg, sig) is False          # 2 keys on 8 devices: empty shards
        for c in caches:
            c.close()
    # one process per GPU: per-rank partial key sums (aggregateAll), added, then the pairing check on the aggregate
    parts = [m.aggregateAll(cache, pks[96 * a:96 * b]) for a, b in ((0, 10000), (10000, 20001), (20001, n))]
    agg = m.p1s_add(cache, parts)
    assert m.verifyAggregate(cache, agg, msg, sig) is True and cache.fetch(4, 576) == gt
    assert m.verifyAggregate(cache, agg, msg, bad) is False
    assert m.verifyAggregate(cache, bytes(144), msg, sig) is False                        # aggregate at infinity


@pytest.mark.gpu
def test_one_message_hash_of_a_32_byte_message(m, cache):
    """k_hash_one takes 32-byte messages through the batch path's prepared-constants expand_message_xmd and every other length through the byte-wise
    absorber: both against the C restatement's hash (the signature is made from it), around the boundary."""
    import c_oracle as co
    sks = [int.from_bytes(hashlib.sha256(b"h32" + bytes([i])).digest(), "little") % o.R or 1 for i in range(5)]
    pks = b"".join(co.sk_to_pk(sk) for sk in sks)
    root = hashlib.sha256(b"a signing root").digest()
    for msg in (root, root[:31], root + b"\x00", b"", root * 3):
        hm = co.hash_to_g2(msg, o.DST_SIG)
        sig, bad = co.g2_mul(hm, sum(sks) % o.R), co.g2_mul(hm, (sum(sks) + 1) % o.R)
        assert m.fastAggregateVerify(cache, pks, msg, sig) is True, len(msg)
        assert m.fastAggregateVerify(cache, pks, msg, bad) is False, len(msg)
    assert m.fastAggregateVerify(cache, pks, root[:31] + bytes([root[31] ^ 1]), co.g2_mul(co.hash_to_g2(root, o.DST_SIG), sum(sks) % o.R)) is False


def test_hash_one_against_rfc9380_and_oracle(m):
    """k_hash_one (two SSWU maps side by side, addition and cofactor clearing on the lane-team engine) through mi355_bls_debug_hash_to_g2: the RFC 9380
    J.10.1 vector of the empty message (tests/test_oracle_kats.py says where it comes from) under the RFC's own DST, and messages of several lengths under
    the scheme's DSTs against the oracle (32 bytes: the prepared-constants path; other lengths: the byte-wise absorber)."""
    import ctypes
    from util import g2_jac_to_affine
    cache = m.BatchedBLSVerifierCache.init(max_sets=64)
    out = ctypes.create_string_buffer(288)

    def h(msg, dst):
        assert m._check(m.lib().mi355_bls_debug_hash_to_g2(cache._h, msg, len(msg), dst, len(dst), out)) == 0
        return g2_jac_to_affine(out.raw)

    rfc = b"QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_"
    assert h(b"", rfc) == ((int("0141ebfbdca40eb85b87142e130ab689c673cf60f1a3e98d69335266f30d9b8d4ac44c1038e9dcdd5393faf5c41fb78a", 16),
                            int("05cb8437535e20ecffaef7752baddf98034139c38452458baeefab379ba13dff5bf5dd71b72418717047f5b0f37da03d", 16)),
                           (int("0503921d7f6a12805e72940b963c0cf3471c7b2a524950ca195d11062ee75ec076daf2d4bc358c4b190c0c98064fdd92", 16),
                            int("12424ac32561493f3fe3c260708a12b7c620e7be00099a974e259ddc7d1f6395c3c811cdd19f1e8dbf3e9ecfdcbab8d6", 16)))
    pop = b"BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"
    for msg, dst in ((b"abc", rfc), (bytes(range(32)), o.DST_SIG), (bytes(range(48)), pop), (b"x" * 200, o.DST_SIG), (bytes(32), pop)):
        assert h(msg, dst) == o.hash_to_g2(msg, dst)
    cache.close()
