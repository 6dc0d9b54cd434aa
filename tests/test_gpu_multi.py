"""GPU tests of the in-library multi-GPU driver (mi355_bls_batch_verify_multi: one host thread, per-device asynchronous
shards, states through pinned host memory, one final exponentiation; bls_batch_verifier.nim:296-371 with devices in
place of threads), the device-resident blob exchange the multi-process bench uses, the cache-less entry point and the
exact blst_p1s_mult_pippenger shape.  Several contexts on ONE device stand in for several devices."""
import hashlib
import random

import pytest

import bls12381_py as o
from util import g1_jac_to_affine, golden

pytestmark = pytest.mark.gpu
RND = hashlib.sha256(b"Mr F was here").digest()


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _case(name):
    c = [x for x in golden("batch")["cases"] if x["name"] == name][0]
    return bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"]), c["n"], c["expect"]


@pytest.mark.parametrize("ngpu", [1, 2, 3, 8])
def test_verify_multi_on_fixtures(m, ngpu):
    """Every golden verdict through ngpu contexts (4 blinding chains, as tests/t_batch_verifier.nim:63); more devices than
    chunks (n = 1, 2, 3 with 8 contexts) leave some shards empty."""
    caches = [m.BatchedBLSVerifierCache.init(max_sets=128, numThreads=4) for _ in range(ngpu)]
    for c in golden("batch")["cases"]:
        rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
        assert m.batchVerifyMulti(caches, rec, rnd) == c["expect"], c["name"]
    assert m.batchVerifyMulti(caches, b"", RND) is False
    for c in caches:
        c.close()


def test_verify_multi_equals_whole_batch(m):
    """1000 distinct tuples, 64 chains, 4 contexts: same verdict and the same final GT value as the single-context call
    and as the C restatement; tampered tuple / infinity public key in one shard -> false."""
    import c_oracle as co
    n, nt = 1000, 64
    rec = co.make_batch(n, seed=555)
    whole = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    assert m.batchVerifyParallel(whole, rec, RND) is True
    gt = whole.fetch(4, 576)
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok and st["gt"] == gt
    caches = [m.BatchedBLSVerifierCache.init(max_sets=300, numThreads=nt) for _ in range(4)]
    assert m.batchVerifyMulti(caches, rec, RND) is True
    assert caches[0].fetch(4, 576) == gt
    # the plan the library uses is the contiguous balanced one
    covered = 0
    for g in range(4):
        lo, hi, first, count = m.shard_plan(n, nt, 4, g)
        assert first == covered and (lo, hi) == (16 * g, 16 * g + 16)
        covered += count
    assert covered == n
    bad = bytearray(rec)
    bad[320 * 700 + 96] ^= 1
    assert m.batchVerifyMulti(caches, bytes(bad), RND) is False
    inf = bytearray(rec)
    inf[320 * 900:320 * 900 + 96] = bytes(96)
    assert m.batchVerifyMulti(caches, bytes(inf), RND) is False
    # device-resident shards
    import torch
    t = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    ptrs = [t.data_ptr() + 320 * m.shard_plan(n, nt, 4, g)[2] for g in range(4)]
    assert m.batchVerifyMulti_device(caches, ptrs, n, RND) is True
    assert caches[0].fetch(4, 576) == gt
    assert m.batchVerifyMulti(caches[:2], rec, RND) is True      # a shard larger than its context (500 > 300): sliced inside the library
    assert caches[0].fetch(4, 576) == gt
    for c in caches + [whole]:
        c.close()


def test_device_resident_blob_exchange(m):
    """What one rank of the multi-process bench does, with the collective replaced by a device-to-device copy: shard
    submit writes (state | ok) into the context's device blob on its stream; the gathered blobs are merged and
    final-exponentiated without a host round trip."""
    import c_oracle as co
    import torch
    n, nt, world = 2048, 256, 4
    rec = co.make_batch(n, seed=4321)
    ok, st = co.batch_verify(rec, RND, nt, stages=True)
    assert ok
    t = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    caches = [m.BatchedBLSVerifierCache.init(max_sets=n // world, numThreads=nt) for _ in range(world)]
    streams = [torch.cuda.Stream() for _ in range(world)]
    gathered = torch.zeros(world * 640, dtype=torch.uint8, device="cuda")
    fv = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=nt)
    mine = [torch.zeros(640, dtype=torch.uint8, device="cuda") for _ in range(world)]
    internal = caches[0].shard_blob_ptr()
    for g in range(world):                                          # every rank's blob goes straight to its send buffer
        caches[g].set_shard_blob_ptr(mine[g].data_ptr())
        assert caches[g].shard_blob_ptr() == mine[g].data_ptr()
    torch.cuda.synchronize()

    def run(buf):
        for g in range(world):
            lo, hi, first, count = m.shard_plan(n, nt, world, g)
            caches[g].shard_submit_device(buf.data_ptr() + 320 * first, n, lo, hi, RND, streams[g].cuda_stream)
        # "all_gather": every rank's blob (written by its submit into its own send buffer) into its slot, on that rank's stream
        for g in range(world):
            with torch.cuda.stream(streams[g]):
                gathered[640 * g:640 * (g + 1)].copy_(mine[g], non_blocking=True)
        for g in range(1, world):
            streams[0].wait_stream(streams[g])
        fv.finalverify_blobs_submit(gathered.data_ptr(), world, 640, streams[0].cuda_stream)
        res = fv.finalverify_wait()
        for g in range(world):
            caches[g].shard_wait()
        return res

    assert run(t) is True
    assert fv.fetch(4, 576) == st["gt"]
    bad = t.clone()
    bad[320 * 1500 + 100] ^= 1
    assert run(bad) is False
    infpk = t.clone()
    infpk[320 * 10:320 * 10 + 96] = 0                              # ok word of shard 0 is 0 -> verdict false
    assert run(infpk) is False
    blob0 = bytes(gathered[:640].cpu().numpy())
    assert blob0[576:580] == bytes(4)
    with pytest.raises(m.BlsGpuError):
        fv.finalverify_wait()                                      # nothing pending
    caches[0].set_shard_blob_ptr(None)
    assert caches[0].shard_blob_ptr() == internal


def test_batch_verify_once(m):
    """Cache-less overloads (bls_batch_verifier.nim:399-416, :475-495): the process-wide default context, regrown on demand."""
    import c_oracle as co
    for name in ("single", "two", "n3", "n17", "wrong_sig", "forged_pair", "inf_pk", "inf_sig", "same_msg_100"):
        rec, rnd, n, expect = _case(name)
        assert m.batchVerifyOnce(rec, rnd, numThreads=4) == expect, name
    assert m.batchVerifyOnce(b"", RND) is False
    rec = co.make_batch(3000, seed=99)                             # larger than the default context's first size
    assert m.batchVerifyOnce(rec, RND) is True
    bad = bytearray(rec)
    bad[320 * 2999 + 97] ^= 2
    assert m.batchVerifyOnce(bytes(bad), RND) is False
    m.lib().mi355_bls_default_ctx_release()


def test_exact_blst_pippenger_shape(m):
    """mi355_p1s_mult_pippenger(ret, points[], npoints, scalars[], nbits, scratch): golden vectors through both list
    conventions; nbits = 64 takes 8-byte scalars (the call `combine` makes, blst_min_pubkey_sig_core.nim:629-636)."""
    for v in golden("msm")["msm"]:
        pts, sc = bytes.fromhex(v["points"]), bytes.fromhex(v["scalars"])
        for per_elem in (False, True):
            out = m.blst_p1s_mult_pippenger(pts, sc, v["nbits"], per_element_pointers=per_elem)
            assert o.g1_to_blst_affine(g1_jac_to_affine(out)).hex() == v["result_affine"], (v["n"], per_elem)
    assert m.blst_p1s_mult_pippenger(b"", b"", 255) == bytes(144)
    v = golden("msm")["msm"][4]                                    # n = 33
    pts, n = bytes.fromhex(v["points"]), v["n"]
    P = [o.g1_from_blst_affine(pts[96 * i:96 * i + 96]) for i in range(n)]
    rng = random.Random(64)
    for nbits in (64, 96, 13):
        sb = (nbits + 7) // 8
        K = [rng.getrandbits(8 * sb) for _ in range(n)]
        sc = b"".join(k.to_bytes(sb, "little") for k in K)
        got = g1_jac_to_affine(m.blst_p1s_mult_pippenger(pts, sc, nbits))
        assert got == o.msm_g1(P, K, nbits), nbits
    m.lib().mi355_bls_default_ctx_release()
