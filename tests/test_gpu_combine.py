"""GPU parity: MultiSignatureSet.combine (bls_batch_verifier.nim:47-106, core :570-647) against the golden
fixture generated from the KAT-pinned oracle; scenarios of tests/t_batch_verifier.nim:139-177."""
import struct

import pytest

import bls12381_py as o
from util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def test_combine_100_same_message(m):
    g = golden("batch")["combine"]
    rnd, n, msg = bytes.fromhex(g["rnd"]), g["n"], bytes.fromhex(g["msg"])
    pks = bytes.fromhex(g["pks"])
    sigs = bytes.fromhex(g["sigs"])
    pk = [pks[96 * i:96 * i + 96] for i in range(n)]
    sg = [sigs[192 * i:192 * i + 192] for i in range(n)]
    cache = m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=4)
    ms = m.MultiSignatureSet.init(pk, msg, sg)
    sigset = ms.combine(cache, rnd)
    scal = struct.unpack("<%dQ" % n, cache.fetch(0, 8 * n))
    assert [str(x) for x in scal] == g["scalars"]                      # word order 3,2,1,0 (core :596-606)
    assert sigset[0].hex() == g["out_pk"] and sigset[2].hex() == g["out_sig"] and sigset[1] == msg
    assert m.batchVerify(cache, [sigset], rnd) is True                 # t_batch_verifier.nim:160-167
    # shuffled (deranged) signatures are rejected (:169-177)
    der = sg[1:] + sg[:1]
    bad = m.MultiSignatureSet.init(pk, msg, der).combine(cache, rnd)
    assert bad[2].hex() == g["deranged_out_sig"]
    assert m.batchVerify(cache, [bad], rnd) is False
    # n == 1 passthrough, add(), n == 0 asserts
    one = m.MultiSignatureSet.init((pk[0], msg, sg[0]))
    assert one.combine(cache, rnd) == (pk[0], msg, sg[0])
    one.add((pk[1], msg, sg[1]))
    two = one.combine(cache, rnd)
    ss = o.combine_scalars(rnd, 2)
    want_pk = o.g1_add(o.g1_mul(o.g1_from_blst_affine(pk[0]), ss[0]), o.g1_mul(o.g1_from_blst_affine(pk[1]), ss[1]))
    assert two[0] == o.g1_to_blst_affine(want_pk)
    with pytest.raises(AssertionError):
        m.MultiSignatureSet.init([], msg, [])
