"""GPU parity: aggregateVerify (bls_sig_min_pubkey.nim:153-199), incl. the reference's forged-pair scenario
(tests/t_batch_verifier.nim:198-244): the forged pair PASSES naive aggregate verification but FAILS batchVerify."""
import pytest

import bls12381_py as o
from util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _case(name):
    c = [x for x in golden("batch")["cases"] if x["name"] == name][0]
    rec = bytes.fromhex(c["sets"])
    n = c["n"]
    pks = [rec[320 * i:320 * i + 96] for i in range(n)]
    msgs = [rec[320 * i + 96:320 * i + 128] for i in range(n)]
    sigs = [o.g2_from_blst_affine(rec[320 * i + 128:320 * i + 320]) for i in range(n)]
    return c, pks, msgs, sigs


def test_aggregate_verify(m):
    cache = m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=4)
    c, pks, msgs, sigs = _case("n9")
    agg = o.g2_to_blst_affine(o.aggregate_g2(sigs))
    assert m.aggregateVerify(cache, pks, msgs, agg) is True
    assert m.aggregateVerify(cache, pks, msgs[1:] + msgs[:1], agg) is False
    assert m.aggregateVerify(cache, pks[:-1], msgs[:-1], agg) is False
    assert m.aggregateVerify(cache, pks, msgs[:-1], agg) is False              # length mismatch (:163-164)
    assert m.aggregateVerify(cache, [], [], agg) is False                      # empty (:165-167)
    assert m.aggregateVerify(cache, [bytes(96)] + pks[1:], msgs, agg) is False  # infinity key
    # forged pair: passes the naive aggregate check, fails the blinded batch check
    c, pks, msgs, sigs = _case("forged_pair")
    aggf = o.g2_to_blst_affine(o.aggregate_g2(sigs))
    assert m.aggregateVerify(cache, pks, msgs, aggf) is True
    assert m.batchVerify(cache, bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])) is False
    # messages of other lengths (the oracle signs them)
    texts = [b"", b"a", b"Mr F was here", bytes(200)]
    keys = [o.keygen_seed(i) for i in range(4)]
    sig = o.aggregate_g2([o.sign(sk, t) for (pk, sk), t in zip(keys, texts)])
    pkb = [o.g1_to_blst_affine(pk) for pk, sk in keys]
    assert m.aggregateVerify(cache, pkb, texts, o.g2_to_blst_affine(sig)) is True
    assert m.aggregateVerify(cache, pkb, [b"x"] + texts[1:], o.g2_to_blst_affine(sig)) is False


def _agg_inputs(m, n, first):
    """n (public key, 32-byte message) pairs with distinct keys and messages + their aggregate signature (device signer; the
    aggregate is the C restatement's sum of the n signatures)."""
    import torch
    import bench
    import c_oracle as co
    dev = torch.device("cuda", 0)
    gen = m.BatchedBLSVerifierCache.init(max_sets=n)
    rec = bytes(bench.sign_records(m, gen, dev, range(first, first + n)).cpu().numpy())
    gen.close()
    pks = [rec[320 * i:320 * i + 96] for i in range(n)]
    msgs = [rec[320 * i + 96:320 * i + 128] for i in range(n)]
    agg = co.g2_sum(b"".join(rec[320 * i + 128:320 * i + 320] for i in range(n)))
    return pks, msgs, agg


@pytest.mark.parametrize("n", [1024, 8193, 20000])
def test_aggregate_verify_at_scale_vs_c_oracle(m, n):
    """1 024 pairs (8 lanes per pair), 8 193 (the first size past the cooperative kernels) and 20 000: verdict AND final GT value
    equal the C restatement of ContextCoreAggregateVerify, in both context modes; one swapped pair of messages and one infinity
    key give false on both sides; a context of 4 096 sets takes the same input in slices."""
    import c_oracle as co
    pks, msgs, agg = _agg_inputs(m, n, 9_000_000 + n)
    want, gt = co.aggregate_verify(pks, msgs, agg, gt=True)
    assert want is True
    swapped = list(msgs)
    swapped[n // 3], swapped[n - 1] = swapped[n - 1], swapped[n // 3]
    wants, gts = co.aggregate_verify(pks, swapped, agg, gt=True)
    assert wants is False
    infk = list(pks)
    infk[n // 2] = bytes(96)
    assert co.aggregate_verify(infk, msgs, agg) is False
    cache = m.BatchedBLSVerifierCache.init(max_sets=n + 8)
    small = m.BatchedBLSVerifierCache.init(max_sets=4096)
    for c in (cache, small):
        for coop in (True, False):
            c.set_cooperative(coop)
            assert m.aggregateVerify(c, pks, msgs, agg) is True
            assert c.fetch(4, 576) == gt
            assert m.aggregateVerify(c, pks, swapped, agg) is False
            assert c.fetch(4, 576) == gts
            assert m.aggregateVerify(c, infk, msgs, agg) is False
    cache.close()
    small.close()


def test_streaming_context(m):
    """ContextCoreAggregateVerify.init / update / finish (blst_min_pubkey_sig_core.nim:321-414) == the one-shot call == the oracle;
    messages of mixed lengths; an infinity key makes update return false and finish false; finish without updates is false; the
    context is reusable after init."""
    import c_oracle as co
    c, pks, msgs, sigs = _case("n9")
    agg = o.g2_to_blst_affine(o.aggregate_g2(sigs))
    cache = m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=4)
    assert m.aggregateVerifyStreaming(cache, pks, msgs, agg) is True
    assert m.aggregateVerifyStreaming(cache, pks, msgs[1:] + msgs[:1], agg) is False
    ctx = m.ContextCoreAggregateVerify(cache)
    ctx.init()
    assert ctx.finish(agg) is False                                  # no pair was added
    ctx.init()
    assert ctx.update(pks[0], msgs[0]) is True
    assert ctx.update(bytes(96), msgs[1]) is False                   # infinity key: BLST_PK_IS_INFINITY
    assert ctx.finish(agg) is False
    texts = [b"", b"a", b"Mr F was here", bytes(200), bytes(range(33))]
    keys = [o.keygen_seed(i) for i in range(5)]
    sig = o.g2_to_blst_affine(o.aggregate_g2([o.sign(sk, t) for (pk, sk), t in zip(keys, texts)]))
    pkb = [o.g1_to_blst_affine(pk) for pk, sk in keys]
    assert co.aggregate_verify(pkb, texts, sig) is True
    assert m.aggregateVerifyStreaming(cache, pkb, texts, sig) is True
    assert m.aggregateVerifyStreaming(cache, pkb, texts[:-1] + [b"?"], sig) is False
    # messages of other lengths through a small context: slices on the general hashing kernel
    tiny = m.BatchedBLSVerifierCache.init(max_sets=2)
    assert m.aggregateVerify(tiny, pkb, texts, sig) is True
    assert m.aggregateVerify(tiny, pkb, [b"x"] + texts[1:], sig) is False
    tiny.close()
    cache.close()


def test_aggregate_all_signatures_and_aggregate_signature_overloads(m):
    """aggregateAll on signatures (core :179-195,211) and finish(signature: AggregateSignature) (core :357): the 288-byte Jacobian
    aggregate the device builds equals the oracle's sum, and aggregateVerify / ContextCoreAggregateVerify.finish accept it."""
    import bench
    import c_oracle as co
    cache = m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=4)
    c, pks, msgs, sigs = _case("n9")
    rec = bytes.fromhex(c["sets"])
    sig_bytes = [rec[320 * i + 128:320 * i + 320] for i in range(c["n"])]
    agg288 = m.aggregateAllSignatures(cache, sig_bytes)
    assert len(agg288) == 288
    want = o.g2_to_blst_affine(o.aggregate_g2(sigs))
    assert bench.g2_jacobian_image_to_affine(agg288) == want
    assert m.aggregateAllSignatures(cache, []) is None
    # an infinity signature in the list contributes nothing; a single signature is itself
    assert bench.g2_jacobian_image_to_affine(m.aggregateAllSignatures(cache, sig_bytes[:4] + [bytes(192)] + sig_bytes[4:])) == want
    assert bench.g2_jacobian_image_to_affine(m.aggregateAllSignatures(cache, sig_bytes[:1])) == sig_bytes[0]
    # the AggregateSignature overloads
    assert m.aggregateVerify(cache, pks, msgs, agg288) is True
    assert m.aggregateVerify(cache, pks, msgs[1:] + msgs[:1], agg288) is False
    ctx = m.ContextCoreAggregateVerify(cache)
    ctx.init()
    for p, x in zip(pks, msgs):
        assert ctx.update(p, x) is True
    assert ctx.finish(agg288) is True
    ctx.init()
    for p, x in zip(pks, msgs[1:] + msgs[:1]):
        ctx.update(p, x)
    assert ctx.finish(agg288) is False
    # more signatures than one wave round, against the C restatement's sum
    pks2, msgs2, agg_aff = _agg_inputs(m, 3000, 7_700_000)
    import torch
    gen = m.BatchedBLSVerifierCache.init(max_sets=3000)
    rec2 = bytes(bench.sign_records(m, gen, torch.device("cuda", 0), range(7_700_000, 7_703_000)).cpu().numpy())
    gen.close()
    a2 = m.aggregateAllSignatures(cache, [rec2[320 * i + 128:320 * i + 320] for i in range(3000)])
    assert bench.g2_jacobian_image_to_affine(a2) == agg_aff
    big = m.BatchedBLSVerifierCache.init(max_sets=4096, numThreads=4)
    assert m.aggregateVerify(big, pks2, msgs2, a2) is True
    assert co.aggregate_verify(pks2, msgs2, agg_aff) is True
