"""Capacity-free batchVerify: the reference cache holds per-thread pairing contexts only and accepts any input.len
(bls_batch_verifier.nim:108-119,141); the device context's workspace is sized for max_sets tuples and larger batches are
processed in slices whose committed states are merged on the device (blst_pairing_merge).  Checked against the C restatement:
identical blinding scalars (the chain of a chunk that a slice boundary cuts is carried over), GT value and verdict."""
import hashlib
import struct

import pytest

pytestmark = pytest.mark.gpu
RND = hashlib.sha256(b"Mr F was here").digest()


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="module")
def batch():
    import c_oracle as co
    n = 3 * 4096 + 17
    return co.make_batch(n, seed=31337), n


@pytest.mark.parametrize("nt", [4096, 4, 1000])
def test_sliced_batch_equals_oracle(m, batch, nt):
    """cap = 4096, n = 3 * 4096 + 17 -> 4 balanced slices.  nt = 4: every chunk (3076 tuples) is cut by slice boundaries; nt = 1000:
    chunks of 12 / 13 tuples, most slice boundaries fall inside a chunk; nt = 4096: 3 / 4 tuples per chunk."""
    import c_oracle as co
    rec, n = batch
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok
    cache = m.BatchedBLSVerifierCache.init(max_sets=4096, numThreads=nt)
    for coop in (True, False):
        cache.set_cooperative(coop)
        assert m.batchVerifyParallel(cache, rec, RND) is True
        assert cache.fetch(4, 576) == st["gt"]
    # the last slice's blinding scalars are the oracle's scalars of those tuples (fetch_stage shows the last slice)
    left, last = n, 0
    for k in range(4, 0, -1):                       # balanced slices: ceil(left / slices left)
        last = (left + k - 1) // k
        left -= last
    assert list(struct.unpack("<%dQ" % last, cache.fetch(0, 8 * last))) == st["r"][n - last:]
    # device-resident records through the asynchronous entry points
    import torch
    d = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    cache.submit_device(d.data_ptr(), n, RND)
    assert cache.wait() is True and cache.fetch(4, 576) == st["gt"]
    # tampered tuple in the third slice; infinity public key in the second
    bad = bytearray(rec)
    bad[320 * 9000 + 96] ^= 1
    okb, stb = co.batch_verify(bytes(bad), RND, nt, stages=True)
    assert not okb
    assert m.batchVerifyParallel(cache, bytes(bad), RND) is False and cache.fetch(4, 576) == stb["gt"]
    inf = bytearray(rec)
    inf[320 * 5000:320 * 5000 + 96] = bytes(96)
    assert co.batch_verify(bytes(inf), RND, nt) is False
    assert m.batchVerifyParallel(cache, bytes(inf), RND) is False
    cache.close()


def test_sliced_serial_chain_and_dispatch(m, batch):
    """batchVerifySerial's single chain over a sliced batch (the host computes the whole chain, every slice uploads its part),
    and batchVerify's dispatch on a small context."""
    import c_oracle as co
    rec, n = batch
    ok, st = co.batch_verify(rec, RND, 0, stages=True)
    assert ok
    cache = m.BatchedBLSVerifierCache.init(max_sets=4096, numThreads=1)
    assert m.batchVerifySerial(cache, rec, RND) is True and cache.fetch(4, 576) == st["gt"]
    assert m.batchVerify(cache, rec, RND) is True                       # numThreads = 1 -> serial path (:440)
    bad = bytearray(rec)
    bad[320 * (n - 1) + 130] ^= 4
    assert m.batchVerifySerial(cache, bytes(bad), RND) is False
    cache.close()


def test_sliced_shards_and_multi(m, batch):
    """Shards larger than their contexts: 3 contexts of 1500 sets for 12 305 tuples (shards of ~4100 -> 3 slices each), the merged
    GT equals the whole-batch GT of the C restatement; the cache-less entry on a default context."""
    import c_oracle as co
    rec, n = batch
    nt = 96
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok
    caches = [m.BatchedBLSVerifierCache.init(max_sets=1500, numThreads=nt) for _ in range(3)]
    assert m.batchVerifyMulti(caches, rec, RND) is True
    assert caches[0].fetch(4, 576) == st["gt"]
    bad = bytearray(rec)
    bad[320 * 12000 + 97] ^= 1
    assert m.batchVerifyMulti(caches, bytes(bad), RND) is False
    for c in caches:
        c.close()


def test_multi_driver_survives_an_enqueue_failure(m):
    """mi355_bls_batch_verify_multi: a failure while enqueuing shard 2 of 3 (injected) is reported, the shards already submitted
    are waited for, and all three contexts are usable right after; a context with a batch still pending is refused BEFORE anything
    is enqueued.  Also prints the host-side start skew between the devices' shards."""
    import ctypes
    import c_oracle as co
    n, nt = 3000, 48
    rec = co.make_batch(n, seed=99)
    caches = [m.BatchedBLSVerifierCache.init(max_sets=1000, numThreads=nt) for _ in range(3)]
    assert m.batchVerifyMulti(caches, rec, RND) is True
    L = m.lib()
    assert L.mi355_bls_debug_fail_next_enqueue(caches[1]._h) == 0
    with pytest.raises(m.BlsGpuError, match="injected"):
        m.batchVerifyMulti(caches, rec, RND)
    assert m.batchVerifyMulti(caches, rec, RND) is True             # nothing left pending on contexts 0 and 2
    for c in caches:
        assert m.batchVerifyParallel(c, rec[:320 * 500], RND) is True
    # a pending batch on context 2: refused in the validation pass, contexts 0 and 1 untouched
    import torch
    d = torch.frombuffer(bytearray(rec[:320 * 800]), dtype=torch.uint8).cuda()
    caches[2].submit_device(d.data_ptr(), 800, RND)
    with pytest.raises(m.BlsGpuError, match="not been waited for"):
        m.batchVerifyMulti(caches, rec, RND)
    assert caches[2].wait() is True
    assert m.batchVerifyMulti(caches, rec, RND) is True
    us = (ctypes.c_float * 8)()
    k = L.mi355_bls_debug_multi_enqueue_us(us, 8)
    print("host-side enqueue times of the shards (us after the call began):", [round(us[i], 1) for i in range(k)])
    assert k == 3 and us[0] <= us[1] <= us[2]
    for c in caches:
        c.close()


def test_staging_grows_for_the_entry_points_around_the_path(m):
    """A context sized for 64 sets takes 500-tuple inputs through every other entry point (their staging buffers grow on demand):
    fromBytes, the wire-format batch verify (growth + slices), the batch signer, combine, fastAggregateVerify - same results as a
    context sized for the input."""
    import c_oracle as co
    import bls12381_py as o
    n = 500
    rec = co.make_batch(n, seed=424242)
    pk, ms, sg = co.compress_sets(rec)
    small = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=16)
    ok, out, st = m.deserializeSets(small, pk, ms, sg)
    assert ok and out == rec and st == bytes(n)
    v, st = m.batchVerifyCompressed(small, pk, ms, sg, RND)
    assert v is True and st == bytes(n)
    okc, stc = co.batch_verify(rec, RND, 16, stages=True)
    assert okc and small.fetch(4, 576) == stc["gt"]
    sgb = bytearray(sg)
    sgb[96 * 400:96 * 401], sgb[96 * 401:96 * 402] = sg[96 * 401:96 * 402], sg[96 * 400:96 * 401]
    v, st = m.batchVerifyCompressed(small, pk, ms, bytes(sgb), RND)
    assert v is False and st == bytes(n)
    # batch signer: the records of a big-enough context
    import hashlib
    sks = [(int.from_bytes(hashlib.sha256(b"grow" + bytes([i % 256, i // 256])).digest(), "little") % (o.R - 1) + 1).to_bytes(32, "little") for i in range(200)]
    msgs = [hashlib.sha256(b"m" + bytes([i % 256])).digest() for i in range(200)]
    big = m.BatchedBLSVerifierCache.init(max_sets=512, numThreads=16)
    small2 = m.BatchedBLSVerifierCache.init(max_sets=8)
    assert m.signSets(small2, sks, msgs) == m.signSets(big, sks, msgs)
    # combine of 300 signatures on one message; fastAggregateVerify of 300 keys
    same, sksum = [], 0
    msg = hashlib.sha256(b"Mr F was here").digest()
    okg, same_rec, _ = m.signSets(big, sks, [msg] * 200)
    assert okg
    pks = [same_rec[320 * i:320 * i + 96] for i in range(200)]
    sigs = [same_rec[320 * i + 128:320 * i + 320] for i in range(200)]
    tiny = m.BatchedBLSVerifierCache.init(max_sets=4)
    assert m.MultiSignatureSet.init(pks, msg, sigs).combine(tiny, RND) == m.MultiSignatureSet.init(pks, msg, sigs).combine(big, RND)
    agg = co.g2_sum(b"".join(sigs))
    assert co.fast_aggregate_verify(b"".join(pks), msg, agg) is True
    assert m.fastAggregateVerify(tiny, pks, msg, agg) is True
    assert m.fastAggregateVerify(tiny, pks[:-1], msg, agg) is False
    for c in (small, small2, big, tiny):
        c.close()


def test_sliced_batches_in_flight_on_several_contexts(m, batch):
    """Three callers (context + stream each), each with a sliced batch in flight at the same time, latency- and throughput-mode
    contexts mixed, batches chained with `after`: every GT value equals the C restatement's, a tampered batch among them is caught."""
    import torch
    import c_oracle as co
    rec, n = batch
    nt = 333
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    bad = bytearray(rec)
    bad[320 * 7777 + 96] ^= 2                      # another message: every point stays on its curve, so the GT values are comparable
    okb, stb = co.batch_verify(bytes(bad), RND, nt, stages=True)
    assert ok and not okb
    d_ok = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    caches = [m.BatchedBLSVerifierCache.init(max_sets=cap, numThreads=nt) for cap in (4096, 3000, 5000)]
    caches[1].set_cooperative(False)
    for rnd_round in range(3):
        plan = [d_ok, d_bad, d_ok] if rnd_round == 1 else [d_ok, d_ok, d_bad] if rnd_round == 2 else [d_ok, d_ok, d_ok]
        for i, (c, s, d) in enumerate(zip(caches, streams, plan)):
            c.submit_device(d.data_ptr(), n, RND, s.cuda_stream, after=caches[i - 1] if i else None)
        for c, d in zip(caches, plan):
            assert c.wait() is (d is d_ok)
            assert c.fetch(4, 576) == (st["gt"] if d is d_ok else stb["gt"])
    for c in caches:
        c.close()


@pytest.mark.parametrize("cap,nt", [(700, 2), (333, 1), (64, 4096), (1, 3)])
def test_chunks_spanning_many_slices(m, cap, nt):
    """A blinding chain that runs through three or more slices (chunk of 1 500 tuples, slices of 600: the carried chain state is read
    and written by the same lane), the serial chain, one-tuple slices (cap = 1: every slice is a single tuple) - GT equal to the C
    restatement's every time."""
    import c_oracle as co
    n = 3000 if cap > 1 else 37
    rec = co.make_batch(n, seed=77000 + cap)
    ok, st = co.batch_verify(rec, RND, nt if nt > 1 else 0, stages=True)
    assert ok
    cache = m.BatchedBLSVerifierCache.init(max_sets=cap, numThreads=nt)
    assert m.batchVerify(cache, rec, RND) is True                   # numThreads = 1 -> the serial chain (:440)
    assert cache.fetch(4, 576) == st["gt"]
    bad = bytearray(rec)
    bad[320 * (n - 2) + 97] ^= 1
    okb, stb = co.batch_verify(bytes(bad), RND, nt if nt > 1 else 0, stages=True)
    assert not okb and m.batchVerify(cache, bytes(bad), RND) is False and cache.fetch(4, 576) == stb["gt"]
    cache.close()
