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
