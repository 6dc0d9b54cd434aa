"""GPU parity: batch signer / input generator (publicFromSecret + coreSign, blst_min_pubkey_sig_core.nim:118-133,
:230-251) against the C restatement and the big-int oracle, and its records through batchVerify."""
import hashlib
import random

import pytest

import bls12381_py as o
import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="module")
def cache(m):
    return m.BatchedBLSVerifierCache.init(max_sets=4096, numThreads=4)


def _msg(i):
    return hashlib.sha256(b"msg" + str(i).encode()).digest()


def test_sign_sets_match_oracles(m, cache):
    rng = random.Random(20260)
    sks = [rng.randrange(1, o.R) for _ in range(130)]
    sks[0], sks[1], sks[2], sks[3] = 1, 2, o.R - 1, (1 << 64) + 5          # edges: tiny, top of range, sparse
    msgs = [_msg(i) for i in range(len(sks))]
    ok, recs, st = m.signSets(cache, [s.to_bytes(32, "little") for s in sks], msgs)
    assert ok and st == bytes(len(sks))
    for i, sk in enumerate(sks):
        rec = recs[320 * i:320 * (i + 1)]
        assert rec[:96] == co.sk_to_pk(sk), i
        assert rec[96:128] == msgs[i]
        assert rec[128:] == co.sign(sk, msgs[i]), i
    # the big-int oracle (pinned to the reference's sk -> pk and proof-of-possession KATs) on a few
    for i in (0, 2, 77):
        pk = o.g1_mul(o.G1_GEN, sks[i])
        sig = o.g2_mul(o.hash_to_g2(msgs[i], o.DST_SIG), sks[i])
        assert recs[320 * i:320 * i + 96] == o.g1_to_blst_affine(pk)
        assert recs[320 * i + 128:320 * (i + 1)] == o.g2_to_blst_affine(sig)
    rnd = hashlib.sha256(b"Mr F was here").digest()
    assert m.batchVerify(cache, recs, rnd) is True


def test_sign_sets_rejects_invalid_keys(m, cache):
    """publicFromSecret returns false for sk == 0 and sk >= r (core :126-129): status 1 and a zeroed record."""
    sks = [5, 0, o.R, o.R + 7, (1 << 256) - 1, 9]
    msgs = [_msg(i) for i in range(len(sks))]
    ok, recs, st = m.signSets(cache, [s.to_bytes(32, "little") for s in sks], msgs)
    assert not ok
    assert st == bytes([0, 1, 1, 1, 1, 0])
    for i in (1, 2, 3, 4):
        assert recs[320 * i:320 * i + 96] == bytes(96) and recs[320 * i + 128:320 * (i + 1)] == bytes(192)
    for i in (0, 5):
        assert recs[320 * i:320 * i + 96] == co.sk_to_pk(sks[i])
        assert recs[320 * i + 128:320 * (i + 1)] == co.sign(sks[i], msgs[i])
    assert m.signSets(cache, [], []) == (True, b"", b"")


def test_sign_sets_device_feeds_batch_verify(m, cache):
    """Device-resident generation -> device-resident verification (the shape bench.py uses), plus a tampered copy."""
    import torch
    n = 1000
    sk = b"".join((int.from_bytes(hashlib.sha256(b"sk" + i.to_bytes(8, "little")).digest(), "little") % (o.R - 1) + 1).to_bytes(32, "little") for i in range(n))
    ms = b"".join(_msg(i) for i in range(n))
    d_sk = torch.frombuffer(bytearray(sk), dtype=torch.uint8).cuda()
    d_ms = torch.frombuffer(bytearray(ms), dtype=torch.uint8).cuda()
    d_out = torch.zeros(320 * n, dtype=torch.uint8, device="cuda")
    ok, st = m.signSets_device(cache, d_sk.data_ptr(), d_ms.data_ptr(), n, d_out.data_ptr())
    assert ok and st == bytes(n)
    rnd = hashlib.sha256(b"Mr F was here").digest()
    assert cache.verify_device(d_out.data_ptr(), n, rnd) is True
    host = bytes(d_out.cpu().numpy())
    assert co.batch_verify(host, rnd, 4) is True
    bad = d_out.clone()
    bad[320 * 500 + 100] ^= 1                      # one message bit
    assert cache.verify_device(bad.data_ptr(), n, rnd) is False
