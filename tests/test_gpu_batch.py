"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the golden fixtures and the oracle.  Scenarios mirror the reference's tests/t_batch_verifier.nim."""
import struct

import pytest

import bls12381_py as o
from util import fp12_from_bytes, fp12_hexlist_to_flat, g1_jac_to_affine, g2_jac_to_affine, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="module")
def cache4(m):
    # Taskpool.new(numThreads = 4) of tests/t_batch_verifier.nim:63
    return m.BatchedBLSVerifierCache.init(max_sets=256, numThreads=4)


def _cases():
    return golden("batch")["cases"]


@pytest.mark.parametrize("name", [c["name"] for c in golden("batch")["cases"]])
def test_verdicts(m, cache4, name):
    c = [x for x in _cases() if x["name"] == name][0]
    rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
    assert m.batchVerify(cache4, rec, rnd) == c["expect"]
    assert m.batchVerifyParallel(cache4, rec, rnd) == c["expect"]
    assert m.batchVerifySerial(cache4, rec, rnd) == c["expect"]
    # cache-less overload shape (tests/t_batch_verifier.nim:75-76): a fresh context
    fresh = m.BatchedBLSVerifierCache.init(max_sets=c["n"], numThreads=4)
    assert m.batchVerify(fresh, rec, rnd) == c["expect"]
    fresh.close()


@pytest.mark.parametrize("name", ["single", "two", "n3", "n9", "n17", "inf_sig", "wrong_sig"])
@pytest.mark.parametrize("mode", ["serial", "chunks4"])
def test_stage_parity(m, cache4, name, mode):
    """Bit-exact canonical stage values: r_i, H(m_i), [r_i]PK_i, sum [r_i]S_i, GT."""
    c = [x for x in _cases() if x["name"] == name][0]
    if mode not in c or "gt" not in c[mode]:
        pytest.skip("no stage fixture")
    rec, rnd, n = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"]), c["n"]
    fn = m.batchVerifySerial if mode == "serial" else m.batchVerifyParallel
    assert fn(cache4, rec, rnd) == c["expect"]
    st = c[mode]
    r = struct.unpack("<%dQ" % n, cache4.fetch(0, 8 * n))
    assert [str(x) for x in r] == st["r"]
    H = cache4.fetch(1, 288 * n)
    P = cache4.fetch(2, 144 * n)
    for i in range(n):
        assert o.g2_to_blst_affine(g2_jac_to_affine(H[288 * i:288 * i + 288])).hex() == st["H"][i]
        assert o.g1_to_blst_affine(g1_jac_to_affine(P[144 * i:144 * i + 144])).hex() == st["rPK"][i]
    assert o.g2_to_blst_affine(g2_jac_to_affine(cache4.fetch(3, 288))).hex() == st["aggsig"]
    assert fp12_from_bytes(cache4.fetch(4, 576)) == fp12_hexlist_to_flat(st["gt"])


def test_empty_and_errors(m, cache4):
    rnd = bytes(32)
    assert m.batchVerify(cache4, b"", rnd) is False               # bls_batch_verifier.nim:137-139
    assert m.batchVerifySerial(cache4, b"", rnd) is False
    with pytest.raises(ValueError):
        m.batchVerify(cache4, bytes(319), rnd)
    big = bytes(320 * 257)
    assert m.batchVerify(cache4, big, rnd) is False                 # beyond max_sets: sliced, not refused (all-zero keys are infinity -> false)


def test_sharded_equals_whole(m):
    """Two 'GPUs' worth of shards on one device: product of shard states == whole-batch state."""
    c = [x for x in _cases() if x["name"] == "n17"][0]
    rec, rnd, n = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"]), c["n"]
    import torch
    cache = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=4)
    t = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    whole = cache.verify_device(t.data_ptr(), n, rnd)
    assert whole is True
    gt_whole = cache.fetch(4, 576)
    states = []
    for lo, hi in ((0, 2), (2, 4)):
        first, count = m.chunk_range(n, 4, lo, hi)
        st, ok = cache.shard_device(t.data_ptr() + 320 * first, n, lo, hi, rnd)
        assert ok
        states.append(st)
    assert cache.finalverify_shards(states) is True
    assert cache.fetch(4, 576) == gt_whole
    # a tampered shard flips the verdict
    bad = bytearray(rec)
    bad[320 * 12 + 100] ^= 0x40
    tb = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    first, count = m.chunk_range(n, 4, 2, 4)
    st_bad, ok = cache.shard_device(tb.data_ptr() + 320 * first, n, 2, 4, rnd)
    assert cache.finalverify_shards([states[0], st_bad]) is False


def test_large_batch_properties(m):
    """4096-tuple batch built by replicating valid fixture tuples (size-independent properties):
    all-valid -> true; one flipped message bit anywhere -> false; order does not matter."""
    c = [x for x in _cases() if x["name"] == "n17"][0]
    rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
    recs = [rec[320 * i:320 * i + 320] for i in range(17)]
    n = 4096
    big = b"".join(recs[i % 17] for i in range(n))
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    assert m.batchVerify(cache, big, rnd) is True
    t = cache.timings()
    assert t["total"] > 0
    bad = bytearray(big)
    bad[320 * 3001 + 96 + 5] ^= 1
    assert m.batchVerify(cache, bytes(bad), rnd) is False
    rev = b"".join(recs[(n - 1 - i) % 17] for i in range(n))
    assert m.batchVerify(cache, rev, rnd) is True


@pytest.mark.parametrize("n,nt", [(1000, 4096), (257, 7), (4096, 4096)])
def test_parity_vs_c_oracle_at_scale(m, n, nt):
    """Distinct random valid tuples (C restatement as generator + checker): every r_i, H(m_i),
    [r_i]PK_i, the aggregated signature and the final GT value are bit-exact after canonicalisation."""
    import c_oracle as co
    rec = co.make_batch(n, seed=77)
    rnd = o.sha256(b"Mr F was here")
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    assert m.batchVerifyParallel(cache, rec, rnd) is True
    ok, st = co.batch_verify(rec, rnd, nt, stages=True)
    assert ok
    r = struct.unpack("<%dQ" % n, cache.fetch(0, 8 * n))
    assert list(r) == st["r"]
    H = cache.fetch(1, 288 * n)
    P = cache.fetch(2, 144 * n)
    step = max(1, n // 64)          # canonicalising needs a python inversion per point: sample
    for i in list(range(0, n, step)) + [n - 1]:
        assert o.g2_to_blst_affine(g2_jac_to_affine(H[288 * i:288 * i + 288])) == st["H"][192 * i:192 * i + 192]
        assert o.g1_to_blst_affine(g1_jac_to_affine(P[144 * i:144 * i + 144])) == st["rPK"][96 * i:96 * i + 96]
    assert o.g2_to_blst_affine(g2_jac_to_affine(cache.fetch(3, 288))) == st["aggsig"]
    assert cache.fetch(4, 576) == st["gt"]
    # one corrupted signature (still a curve point: swap two signatures) -> false on both
    bad = bytearray(rec)
    bad[128:320], bad[320 + 128:640] = rec[320 + 128:640], rec[128:320]
    assert m.batchVerifyParallel(cache, bytes(bad), rnd) is False
    assert co.batch_verify(bytes(bad), rnd, nt) is False


def test_cooperative_and_plain_kernels_agree(m):
    """Batches that do not fill the chip run hashing and Miller lines on the lane-team engine by default (16 lanes per set, csrc/teamvm.hpp);
    the one-lane-per-set kernels must give the same H(m_i) points, GT value and verdict."""
    import c_oracle as co
    n = 600
    rec = co.make_batch(n, seed=31337)
    rnd = o.sha256(b"Mr F was here")
    a = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=64)
    b = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=64)
    b.set_cooperative(False)
    assert m.batchVerify(a, rec, rnd) is True and m.batchVerify(b, rec, rnd) is True
    assert a.fetch(4, 576) == b.fetch(4, 576)
    Ha, Hb = a.fetch(1, 288 * n), b.fetch(1, 288 * n)
    for i in (0, 1, 299, n - 1):
        assert g2_jac_to_affine(Ha[288 * i:288 * i + 288]) == g2_jac_to_affine(Hb[288 * i:288 * i + 288])
    ok, st = co.batch_verify(rec, rnd, 64, stages=True)
    assert ok and st["gt"] == a.fetch(4, 576)
    # the fold of the line products: a call enqueued while nothing else is in flight uses the Fp12 engine (k_fold) in either mode; with another
    # context's batch pending the one-lane-per-set context takes k_lineprod2.  Same GT bytes.
    import torch
    d = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    other = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=64)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    other.submit_device(d.data_ptr(), n, rnd, s.cuda_stream)
    assert b.verify_device(d.data_ptr(), n, rnd) is True
    assert other.wait() is True
    assert b.fetch(4, 576) == st["gt"] and other.fetch(4, 576) == st["gt"]
    # the choice is recorded per call (mi355_bls_last_fold_form): `other` was enqueued with nothing in flight -> engine fold; `b` (the
    # one-lane-per-set context) behind it -> k_lineprod2; a latency-mode context always folds on the engine
    assert other.fold_form() == 1 and b.fold_form() == 0 and a.fold_form() == 1
    for c in (a, b, other):
        c.close()


def test_submit_wait(m):
    """Asynchronous entry points: several contexts kept in flight from one host thread give the verdicts of the
    blocking call; a context takes one batch at a time."""
    import torch
    c = [x for x in _cases() if x["name"] == "n17"][0]
    rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
    n = 17
    bad = bytearray(rec)
    bad[320 * 5 + 100] ^= 1
    d_ok = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
    d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    caches = [m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=4) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    plan = [d_ok, d_bad, d_ok, d_ok, d_bad, d_bad, d_ok]
    got, busy = [], [None] * 3
    for i, d in enumerate(plan):
        s = i % 3
        if busy[s] is not None:
            got.append((busy[s], caches[s].wait()))
        caches[s].submit_device(d.data_ptr(), n, rnd, streams[s].cuda_stream, after=caches[(s - 1) % 3] if i % 2 else None)
        busy[s] = i
    with pytest.raises(m.BlsGpuError):
        caches[0].submit_device(d_ok.data_ptr(), n, rnd, streams[0].cuda_stream)      # still pending
    for j in range(3):
        s = (len(plan) + j) % 3
        got.append((busy[s], caches[s].wait()))
    assert dict(got) == {i: (d is d_ok) for i, d in enumerate(plan)}
    with pytest.raises(m.BlsGpuError):
        caches[0].wait()                                                              # nothing pending
    # the process-wide count of batches in flight (it picks the fold of the line products) stays balanced: submit +1, wait -1,
    # a refused submit 0, a context destroyed with its batch pending -1
    L = m.lib()
    assert L.mi355_bls_debug_batches_in_flight() == 0
    caches[1].submit_device(d_ok.data_ptr(), n, rnd, streams[1].cuda_stream)
    caches[2].submit_device(d_bad.data_ptr(), n, rnd, streams[2].cuda_stream)
    assert L.mi355_bls_debug_batches_in_flight() == 2
    with pytest.raises(m.BlsGpuError):
        caches[1].submit_device(d_ok.data_ptr(), n, rnd, streams[1].cuda_stream)
    assert L.mi355_bls_debug_batches_in_flight() == 2
    assert caches[1].wait() is True
    assert L.mi355_bls_debug_batches_in_flight() == 1
    torch.cuda.synchronize()
    caches[2].close()                                                                 # never waited for
    assert L.mi355_bls_debug_batches_in_flight() == 0
    assert caches[0].verify_device(d_ok.data_ptr(), n, rnd) is True


@pytest.mark.parametrize("n", [1024, 2048, 39999, 40000, 40960])
def test_bucket_fold_of_the_signature_side(m, n):
    """n >= 64 (SIG_BUCKET_MIN): the signatures are folded into digit buckets that become extra Miller pairs (4-bit digits below
    40 000 tuples, 8-bit from there).  sum [r_i]S_i (folded from the buckets on demand) and the final GT value must
    still be the C restatement's, also with infinity signatures in the batch (they contribute nothing)."""
    import c_oracle as co
    rec = bytearray(co.make_batch(n, seed=4242))
    rnd = o.sha256(b"Mr F was here")
    nt = 4096
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nt)
    assert m.batchVerifyParallel(cache, bytes(rec), rnd) is True
    ok, st = co.batch_verify(bytes(rec), rnd, nt, stages=True)
    assert ok
    assert o.g2_to_blst_affine(g2_jac_to_affine(cache.fetch(3, 288))) == st["aggsig"]
    assert cache.fetch(4, 576) == st["gt"]
    for i in (0, 777, n - 1):                                   # infinity signatures: all-zero affine image
        rec[320 * i + 128:320 * i + 320] = bytes(192)
    assert m.batchVerifyParallel(cache, bytes(rec), rnd) is False
    ok, st = co.batch_verify(bytes(rec), rnd, nt, stages=True)
    assert not ok
    assert o.g2_to_blst_affine(g2_jac_to_affine(cache.fetch(3, 288))) == st["aggsig"]
    assert cache.fetch(4, 576) == st["gt"]
    cache.close()


@pytest.mark.parametrize("n", [63, 64, 223, 224, 225, 447, 448, 449, 1792, 1793])
def test_hand_over_sizes_of_the_row_executor(m, n):
    """Around the sizes where the latency path changes executor (csrc/rowvm.hpp: a workgroup of four waves per message / pair, one per CU up to 224 items, two
    per CU up to 448, the lane-team engine above; k_hash_map_rows: a (message, u) pair per DPP row up to 1 792 messages; the signature side's buckets become extra pairs from
    64 sets): the verdict AND the GT value of the C restatement, valid batch and one with a wrong signature."""
    import c_oracle as co
    rec = bytearray(co.make_batch(n, seed=777 + n))
    rnd = o.sha256(b"Mr F was here")
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=7)
    assert m.batchVerifyParallel(cache, bytes(rec), rnd) is True
    ok, st = co.batch_verify(bytes(rec), rnd, 7, stages=True)
    assert ok and cache.fetch(4, 576) == st["gt"]
    rec[320 * (n // 2) + 128:320 * (n // 2) + 320], rec[128:320] = rec[128:320], rec[320 * (n // 2) + 128:320 * (n // 2) + 320]     # two signatures swapped
    assert m.batchVerifyParallel(cache, bytes(rec), rnd) is False
    ok, st = co.batch_verify(bytes(rec), rnd, 7, stages=True)
    assert not ok and cache.fetch(4, 576) == st["gt"]
    cache.close()


@pytest.mark.parametrize("n", [65537, 100003, 131072])
def test_sizes_beyond_one_wave_per_simd(m, n):
    """More than 64 x 1024 tuples (several rounds of waves, side-stream path on/off): all-valid -> true,
    one swapped signature anywhere -> false.  Tuples are distinct for the first 8192, then tiled."""
    import c_oracle as co
    base = co.make_batch(8192, seed=123)
    rec = (base * (n // 8192 + 1))[:320 * n]
    rnd = o.sha256(b"Mr F was here")
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    assert m.batchVerify(cache, rec, rnd) is True
    bad = bytearray(rec)
    i, j = n - 1, n - 2
    bad[320 * i + 128:320 * i + 320], bad[320 * j + 128:320 * j + 320] = rec[320 * j + 128:320 * j + 320], rec[320 * i + 128:320 * i + 320]
    if rec[320 * i + 128:320 * i + 320] != rec[320 * j + 128:320 * j + 320]:
        assert m.batchVerify(cache, bytes(bad), rnd) is False
    cache.close()


def test_randomised_batches_against_the_c_oracle(m):
    """Soak: 40 batches of random size (1 .. 700) on ONE reused context, each with a random defect or none - duplicated tuples
    (equal points meet in a signature bucket: the doubling case of the mixed addition), an all-zero signature or public key, a
    flipped message / key / signature bit, signatures swapped between tuples - both modes, random numThreads: verdict and, for
    batches the oracle accepts or rejects alike, the GT value equal to the C restatement's."""
    import random
    import c_oracle as co
    rng = random.Random(20261003)
    caches = {True: m.BatchedBLSVerifierCache.init(max_sets=700, numThreads=64), False: m.BatchedBLSVerifierCache.init(max_sets=700, numThreads=64)}
    caches[False].set_cooperative(False)
    pool = co.make_batch(700, seed=424242)
    for it in range(40):
        n = rng.choice([1, 2, 3, 5, 8, 63, 64, 65, 100, 257, 700, rng.randrange(1, 700)])
        idx = [rng.randrange(700) for _ in range(n)] if it % 3 == 0 else rng.sample(range(700), n)     # with / without duplicates
        rec = bytearray(b"".join(pool[320 * i:320 * i + 320] for i in idx))
        defect = rng.choice(["none", "none", "msg", "pk", "sig", "swap", "inf_sig", "inf_pk"])
        t = rng.randrange(n)
        if defect == "msg":
            rec[320 * t + 96 + rng.randrange(32)] ^= 1 << rng.randrange(8)
        elif defect == "pk" and n > 1:
            u = (t + 1) % n
            rec[320 * t:320 * t + 96] = rec[320 * u:320 * u + 96]
        elif defect == "sig" and n > 1:
            u = (t + 1) % n
            rec[320 * t + 128:320 * t + 320] = rec[320 * u + 128:320 * u + 320]
        elif defect == "swap" and n > 1:
            u = (t + 1) % n
            a, b = bytes(rec[320 * t + 128:320 * t + 320]), bytes(rec[320 * u + 128:320 * u + 320])
            rec[320 * t + 128:320 * t + 320], rec[320 * u + 128:320 * u + 320] = b, a
        elif defect == "inf_sig":
            rec[320 * t + 128:320 * t + 320] = bytes(192)
        elif defect == "inf_pk":
            rec[320 * t:320 * t + 96] = bytes(96)
        rec = bytes(rec)
        rnd = o.sha256(b"soak" + bytes([it]))
        nt = rng.choice([1, 4, 64, 4096])
        want, st = co.batch_verify(rec, rnd, nt if (nt > 1 and n >= 3) else 0, stages=True)       # batchVerify's own choice of the serial chain
        for coop, cache in caches.items():
            cache.numThreads = nt
            m._check(m.lib().mi355_bls_ctx_set_num_threads(cache._h, nt))
            got = m.batchVerify(cache, rec, rnd)
            assert got == want, (it, n, defect, nt, coop)
            if defect != "inf_pk":
                assert cache.fetch(4, 576) == st["gt"], (it, n, defect, nt, coop)
    for c in caches.values():
        c.close()


def test_keys_outside_g1_take_the_complete_formulas(m):
    """batchVerify is handed validated keys, but the kernel must not depend on it: k_pkmul's assembly loop uses incomplete additions with ONE zero test
    (Z3 == 0 -> the lane is recomputed with the complete formulas).  Keys of order 3 ((0, +-2): their window table holds the point at infinity) and keys of
    full order h * r (on the curve, outside G1) among ordinary ones, in both modes and beyond one wave: [r_i]PK_i equal to the oracle's scalar
    multiplication for EVERY tuple, verdict and GT equal to the C restatement's."""
    import c_oracle as co
    n = 150
    rec = bytearray(co.make_batch(n, seed=777))
    odd = {}
    odd[0] = (0, 2)
    odd[1] = (0, o.P - 2)
    odd[64] = (0, 2)
    x = 1
    for slot in (2, 63, 65, 129, 149):
        while True:
            x += 1
            y = o.fp_sqrt((x * x * x + 4) % o.P)
            if y is not None and not o.g1_in_subgroup((x, y)):
                break
        odd[slot] = (x, y)
    for slot, pt in odd.items():
        assert o.g1_on_curve(pt)
        rec[320 * slot:320 * slot + 96] = o.g1_to_blst_affine(pt)
    rec = bytes(rec)
    rnd = o.sha256(b"outside G1")
    for coop in (True, False):
        cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=4)
        cache.set_cooperative(coop)
        want, st = co.batch_verify(rec, rnd, 4, stages=True)
        assert m.batchVerify(cache, rec, rnd) == want
        r = struct.unpack("<%dQ" % n, cache.fetch(0, 8 * n))
        P = cache.fetch(2, 144 * n)
        for i in range(n):
            pk = o.g1_from_blst_affine(rec[320 * i:320 * i + 96])
            assert g1_jac_to_affine(P[144 * i:144 * i + 144]) == o.g1_mul(pk, r[i]), (i, coop)
        assert cache.fetch(4, 576) == st["gt"]
        cache.close()
