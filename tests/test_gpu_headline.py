"""GPU parity at and above the headline batch size (BASELINE.json: 65 536-tuple batch; config 5: 2^20 tuples in
8 shards of 131 072).  Distinct tuples come from the device signer (parity-tested in test_gpu_sign.py); the checker
is the C restatement of the reference algorithm (oracle/bls_oracle.c, OpenMP) and, for the blinding chains of the
2^20 batch, the big-int oracle.  Reference: bls_batch_verifier.nim:296-371 (chunks, merge, finalVerify)."""
import hashlib
import struct

import pytest

import bls12381_py as o
from util import g1_jac_to_affine, g2_jac_to_affine

pytestmark = pytest.mark.gpu

RND = hashlib.sha256(b"Mr F was here").digest()


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _signed_records(m, n, first=0):
    """n distinct valid tuples resident in HBM (device signer, 65 536 per call)."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    gen = m.BatchedBLSVerifierCache.init(max_sets=min(n, 65536))
    parts = [bench.sign_records(m, gen, dev, range(first + s, first + min(s + 65536, n))) for s in range(0, n, 65536)]
    gen.close()
    return parts[0] if len(parts) == 1 else torch.cat(parts)


@pytest.mark.parametrize("n", [65536, 131072])
def test_stage_parity_at_headline_sizes(m, n):
    """Every r_i, a sample of H(m_i) and [r_i]PK_i, sum [r_i]S_i and the final GT value are bit-exact against the C
    restatement at 65 536 tuples (one wave per SIMD, 8-bit signature buckets) and 131 072 (two rounds of waves in
    every per-tuple kernel, m > 1 in the line products).  One swapped pair of signatures -> false on both sides."""
    import c_oracle as co
    d = _signed_records(m, n, first=7_000_000)
    rec = bytes(d.cpu().numpy())
    nt = 4096
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    assert cache.verify_device(d.data_ptr(), n, RND) is True
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok
    assert list(struct.unpack("<%dQ" % n, cache.fetch(0, 8 * n))) == st["r"]
    H, P = cache.fetch(1, 288 * n), cache.fetch(2, 144 * n)
    for i in list(range(0, n, n // 24)) + [63, 64, 65535, n - 1]:
        assert o.g2_to_blst_affine(g2_jac_to_affine(H[288 * i:288 * i + 288])) == st["H"][192 * i:192 * i + 192], i
        assert o.g1_to_blst_affine(g1_jac_to_affine(P[144 * i:144 * i + 144])) == st["rPK"][96 * i:96 * i + 96], i
    assert o.g2_to_blst_affine(g2_jac_to_affine(cache.fetch(3, 288))) == st["aggsig"]
    assert cache.fetch(4, 576) == st["gt"]
    # throughput mode (the contexts of bench.py's timed region: one lane per item, no fork stream, k_lineprod2 instead of
    # the engine fold, two-wave tail): the same GT value and aggregate
    tp = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    tp.set_cooperative(False)
    assert tp.verify_device(d.data_ptr(), n, RND) is True
    assert tp.fetch(4, 576) == st["gt"]
    assert o.g2_to_blst_affine(g2_jac_to_affine(tp.fetch(3, 288))) == st["aggsig"]
    # ... and with another context's batch in flight, as in the timed region (a lone caller folds on the engine in either mode)
    import torch
    other = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=4)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    other.submit_device(d.data_ptr(), 64, RND, side.cuda_stream)
    assert tp.verify_device(d.data_ptr(), n, RND) is True
    assert other.wait() is True
    assert tp.fetch(4, 576) == st["gt"]
    other.close()
    tp.close()
    bad = d.clone()
    i, j = n - 1, n // 2
    bad[320 * i + 128:320 * i + 320] = d[320 * j + 128:320 * j + 320]
    bad[320 * j + 128:320 * j + 320] = d[320 * i + 128:320 * i + 320]
    assert cache.verify_device(bad.data_ptr(), n, RND) is False
    ok_bad, st_bad = co.batch_verify(bytes(bad.cpu().numpy()), RND, nt, stages=True)
    assert not ok_bad
    assert cache.fetch(4, 576) == st_bad["gt"]
    cache.close()


def test_config5_emulation_2pow20_in_8_shards(m):
    """BASELINE config 5 on one device: 2^20 distinct tuples, 8 x 4096 blinding chains, 8 contiguous chunk blocks of
    131 072 tuples through mi355_bls_batch_shard_device (the per-GPU work of bls_batch_verifier.nim:326-357), merged
    by mi355_bls_finalverify_shards (:360-371).  The merged GT value equals the whole-batch call's, the blinding
    scalars of every shard are the oracle's, and a tampered shard flips the verdict."""
    import torch
    n, world, nt = 1 << 20, 8, 8 * 4096
    d = _signed_records(m, n, first=9_000_000)
    whole = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    assert whole.verify_device(d.data_ptr(), n, RND) is True
    gt_whole = whole.fetch(4, 576)
    r_whole = struct.unpack("<%dQ" % n, whole.fetch(0, 8 * n))
    whole.close()
    # the C restatement of the reference on the WHOLE 2^20-tuple batch (OpenMP, about two minutes on the GPU box's host cores): verdict
    # and final GT value.  (Rounds 1-3 compared the device with itself at this size; per-shard arithmetic was oracle-pinned at 131 072.)
    import os
    if (os.cpu_count() or 1) >= 8:
        import c_oracle as co
        ok_c, st_c = co.batch_verify(bytes(d.cpu().numpy()), RND, nt, stages=True)
        assert ok_c is True and st_c["gt"] == gt_whole
        del st_c
    # the oracle's chains for a few chunks of every shard (chunk c covers 32 tuples here)
    per = n // nt
    for c in (0, 1, 4095, 4096, 20000, nt - 1):
        seed = o.blinding_seed(RND, c)
        for j in range(per):
            seed, r = o.blinding_next(seed)
            assert r_whole[c * per + j] == r, (c, j)
    shard = m.BatchedBLSVerifierCache.init(max_sets=n // world, numThreads=nt)
    states = []
    for g in range(world):
        lo, hi = g * (nt // world), (g + 1) * (nt // world)
        first, count = m.chunk_range(n, nt, lo, hi)
        assert (first, count) == (g * (n // world), n // world)
        st, ok = shard.shard_device(d.data_ptr() + 320 * first, n, lo, hi, RND)
        assert ok
        r_sh = struct.unpack("<%dQ" % count, shard.fetch(0, 8 * count))
        assert r_sh == r_whole[first:first + count]
        states.append(st)
    fv = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=nt)
    assert fv.finalverify_shards(states) is True
    assert fv.fetch(4, 576) == gt_whole
    # tamper one message bit in shard 5 -> its state changes, the merged verdict is false
    g = 5
    first = g * (n // world)
    bad = d[320 * first:320 * (first + n // world)].clone()
    bad[320 * 77777 + 96 + 3] ^= 0x10
    st_bad, ok = shard.shard_device(bad.data_ptr(), n, g * (nt // world), (g + 1) * (nt // world), RND)
    assert ok and st_bad != states[g]
    assert fv.finalverify_shards(states[:g] + [st_bad] + states[g + 1:]) is False
    shard.close()
    fv.close()


@pytest.mark.parametrize("n", [262144])
def test_one_blocking_call_above_the_headline(m, n):
    """(was tests/gpu_probe_big.py) one blocking call on 4 x 65 536 distinct tuples: true; an infinity public key
    anywhere -> false (BLST_PK_IS_INFINITY, blst_min_pubkey_sig_core.nim:559-567)."""
    d = _signed_records(m, n, first=11_000_000)
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    assert cache.verify_device(d.data_ptr(), n, RND) is True
    t = cache.timings()
    assert 0 < t["total"] < 500
    bad = d.clone()
    bad[320 * 200001:320 * 200001 + 96] = 0
    assert cache.verify_device(bad.data_ptr(), n, RND) is False
    cache.close()


@pytest.mark.parametrize("n", [8192, 8193, 16384, 16385, 39999, 40000])
def test_gt_parity_at_the_path_boundaries(m, n):
    """The sizes where run_shard changes path: 8 192 / 8 193 (8 lanes per set -> one lane per set), 16 384 / 16 385 ([r]PK and
    the signature side on the fork stream -> only the signature side and its extra pairs' lines), 39 999 / 40 000 (4-bit ->
    8-bit signature buckets).  Latency and throughput mode: GT and aggregate bit-exact against the C restatement, one
    context reused across a true and a false batch."""
    import c_oracle as co
    d = _signed_records(m, n, first=13_000_000)
    rec = bytes(d.cpu().numpy())
    nt = 4096
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok
    bad = d.clone()
    bad[320 * (n - 1) + 96] ^= 1                                        # last message: the signature no longer matches
    for coop in (True, False):
        cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
        cache.set_cooperative(coop)
        assert cache.verify_device(d.data_ptr(), n, RND) is True
        assert cache.fetch(4, 576) == st["gt"], coop
        assert o.g2_to_blst_affine(g2_jac_to_affine(cache.fetch(3, 288))) == st["aggsig"]
        assert cache.verify_device(bad.data_ptr(), n, RND) is False
        assert cache.verify_device(d.data_ptr(), n, RND) is True
        assert cache.fetch(4, 576) == st["gt"]
        cache.close()
