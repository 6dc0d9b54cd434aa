"""Many independent batches in one device pass (mi355_bls_batch_verify_many): every verdict equals that of a separate
batchVerify call (bls_batch_verifier.nim:420-495) and of the C restatement on that batch; the merged pass uses each batch's own
blinding scalars (its chain partition and random bytes), checked against the oracle's r_i."""
import hashlib
import struct

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _rnd(i):
    return hashlib.sha256(b"many" + bytes([i])).digest()


def test_many_batches_verdicts_and_scalars(m):
    import c_oracle as co
    sizes = [100, 1, 2, 3, 0, 257, 64, 1000]
    nt = 7
    batches = [co.make_batch(n, seed=900 + 17 * i) if n else b"" for i, n in enumerate(sizes)]
    rnds = [_rnd(i) for i in range(len(sizes))]
    cache = m.BatchedBLSVerifierCache.init(max_sets=2048, numThreads=nt)
    want = [bool(n) for n in sizes]                                   # all valid; the empty batch is false
    assert m.batchVerifyMany(cache, batches, rnds) == want
    # the scalars of the merged pass are each batch's own: the oracle's r_i of that batch alone (serial chain for n < 3)
    total = sum(sizes)
    r = list(struct.unpack("<%dQ" % total, cache.fetch(0, 8 * total)))
    off = 0
    for n, rec, rnd in zip(sizes, batches, rnds):
        if n:
            ok, st = co.batch_verify(rec, rnd, nt if n >= 3 else 0, stages=True)
            assert ok and r[off:off + n] == st["r"], n
        off += n
    # one tampered batch (another message in batch 5), one with an infinity key (batch 0): exactly those are false
    bad = list(batches)
    b5 = bytearray(batches[5])
    b5[320 * 200 + 96] ^= 1
    bad[5] = bytes(b5)
    b0 = bytearray(batches[0])
    b0[320 * 50:320 * 50 + 96] = bytes(96)
    bad[0] = bytes(b0)
    got = m.batchVerifyMany(cache, bad, rnds)
    assert got == [False, True, True, True, False, False, True, True]
    assert got == [m.batchVerify(cache, b, rd) if b else False for b, rd in zip(bad, rnds)]
    assert [co.batch_verify(b, rd, nt if len(b) // 320 >= 3 else 0) if b else False for b, rd in zip(bad, rnds)] == got
    # a union larger than the context: verified one by one, same verdicts; no batches at all; all empty
    small = m.BatchedBLSVerifierCache.init(max_sets=300, numThreads=nt)
    assert m.batchVerifyMany(small, bad, rnds) == got
    assert m.batchVerifyMany(small, batches, rnds) == want
    assert m.batchVerifyMany(cache, [], []) == []
    assert m.batchVerifyMany(cache, [b"", b""], rnds[:2]) == [False, False]
    # device-resident tuples
    import torch
    d = torch.frombuffer(bytearray(b"".join(batches)), dtype=torch.uint8).cuda()
    assert m.batchVerifyMany_device(cache, d.data_ptr(), sizes, rnds) == want
    cache.close()
    small.close()


def test_sixteen_batches_of_4096(m):
    """BASELINE config 2 shape, sixteen at a time: 16 x 4 096 distinct tuples in one pass on a 65 536-set context; one forged batch
    is singled out."""
    import torch
    import bench
    n, k = 4096, 16
    dev = torch.device("cuda", 0)
    cache = m.BatchedBLSVerifierCache.init(max_sets=n * k)
    d = bench.sign_records(m, cache, dev, range(3_000_000, 3_000_000 + n * k))
    rnds = [_rnd(100 + i) for i in range(k)]
    assert m.batchVerifyMany_device(cache, d.data_ptr(), [n] * k, rnds) == [True] * k
    print("16 x 4096 in one pass, timings(ms):", cache.timings())
    bad = d.clone()
    i, j = 9 * n + 5, 9 * n + 4000                                   # two signatures of batch 9 swapped
    bad[320 * i + 128:320 * i + 320] = d[320 * j + 128:320 * j + 320]
    bad[320 * j + 128:320 * j + 320] = d[320 * i + 128:320 * i + 320]
    assert m.batchVerifyMany_device(cache, bad.data_ptr(), [n] * k, rnds) == [b != 9 for b in range(k)]
    cache.close()


def test_same_rnd_in_two_batches_cannot_cancel_across_them(m):
    """Two batches with the SAME secureRandomBytes and the same count have identical blinding scalars at identical indices, so a
    forger can make errors cancel ACROSS them: sig_A[i] + D in one, sig_B[i] - D in the other.  The product of the two batch checks
    is then one although both batches are invalid; k separate batchVerify calls reject both.  The library must notice the equal
    rnds, skip the merged pass and return the verdicts of separate calls."""
    import bls12381_py as o
    import c_oracle as co
    n, nt, i = 8, 4, 5
    a = bytearray(co.make_batch(n, seed=4100))
    b = bytearray(co.make_batch(n, seed=4200))
    d = o.g2_from_blst_affine(bytes(a[320 * 2 + 128:320 * 2 + 320]))                     # any G2 point
    sa = o.g2_from_blst_affine(bytes(a[320 * i + 128:320 * i + 320]))
    sb = o.g2_from_blst_affine(bytes(b[320 * i + 128:320 * i + 320]))
    a[320 * i + 128:320 * i + 320] = o.g2_to_blst_affine(o.g2_add(sa, d))
    b[320 * i + 128:320 * i + 320] = o.g2_to_blst_affine(o.g2_add(sb, o.g2_neg(d)))
    a, b = bytes(a), bytes(b)
    rnd = _rnd(77)
    cache = m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=nt)
    assert co.batch_verify(a, rnd, nt) is False and co.batch_verify(b, rnd, nt) is False
    assert m.batchVerify(cache, a, rnd) is False and m.batchVerify(cache, b, rnd) is False
    assert m.batchVerifyMany(cache, [a, b], [rnd, rnd]) == [False, False]                 # equal rnds: verified one by one
    # with independent rnds the merged pass runs and rejects too (the scalars at index i differ, nothing cancels)
    assert m.batchVerifyMany(cache, [a, b], [rnd, _rnd(78)]) == [False, False]
    # equal rnds on VALID batches: still correct, just not merged
    va, vb = co.make_batch(n, seed=4100), co.make_batch(n, seed=4200)
    assert m.batchVerifyMany(cache, [va, vb, b""], [rnd, rnd, rnd]) == [True, True, False]
