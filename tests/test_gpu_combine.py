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


@pytest.mark.parametrize("n", [2, 33, 4096])
def test_combine_is_two_pippenger_runs(m, n):
    """combine = blst_p1s_mult_pippenger + blst_p2s_mult_pippenger with 64-bit scalars (core :629-646): scalars, combined key
    and combined signature byte-exact against the C restatement at sizes where the bucket kernels really bucket; the
    combined set verifies, a deranged one does not."""
    import c_oracle as co
    msg = o.sha256(b"same message")
    sks = [1000 + 7 * i for i in range(n)]
    base_pk = [co.sk_to_pk(s) for s in sks[:64]]
    h = co.hash_to_g2(msg, o.DST_SIG)
    base_sg = [co.g2_mul(h, s) for s in sks[:64]]
    pks = [base_pk[i % 64] for i in range(n)]
    sgs = [base_sg[i % 64] for i in range(n)]
    rnd = o.sha256(b"combine rnd")
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=4)
    out = m.MultiSignatureSet.init(pks, msg, sgs).combine(cache, rnd)
    want_pk, want_sg, want_sc = co.combine(rnd, b"".join(pks), b"".join(sgs))
    assert list(struct.unpack("<%dQ" % n, cache.fetch(0, 8 * n))) == want_sc
    assert out[0] == want_pk and out[2] == want_sg
    assert m.batchVerify(cache, [out], rnd) is True
    bad = m.MultiSignatureSet.init(pks, msg, sgs[1:] + sgs[:1]).combine(cache, rnd)
    assert m.batchVerify(cache, [bad], rnd) is False
    print("combine", n, cache.timings())


def test_g2_pippenger_entry_points(m):
    """mi355_p2s_mult_pippenger (exact blst shape, both list conventions, nbits 64 / 255 / odd) and the context / device forms
    against the C restatement's naive G2 MSM."""
    import random
    import c_oracle as co
    import torch
    from util import g2_jac_to_affine
    rng = random.Random(2)
    h = co.hash_to_g2(b"g2 msm", o.DST_SIG)
    base = [co.g2_mul(h, rng.randrange(1, o.R)) for _ in range(40)]
    for n in (1, 2, 40, 1000):
        pts = b"".join(base[i % 40] for i in range(n))
        for nbits in (255, 64, 21):
            sb = (nbits + 7) // 8
            sc = bytes(rng.getrandbits(8) for _ in range(sb * n))
            want = co.msm_g2(pts, sc, nbits, sb)
            for per_elem in ((False, True) if n <= 40 else (False,)):
                got = m.blst_p2s_mult_pippenger(pts, sc, nbits, per_element_pointers=per_elem)
                assert o.g2_to_blst_affine(g2_jac_to_affine(got)) == want, (n, nbits, per_elem)
    assert m.blst_p2s_mult_pippenger(b"", b"", 255) == bytes(288)
    # infinity in the list, cancellation
    q = o.g2_from_blst_affine(base[0])
    pts = base[0] + bytes(192) + o.g2_to_blst_affine(o.g2_neg(q))
    sc = (5).to_bytes(32, "little") * 3
    assert g2_jac_to_affine(m.blst_p2s_mult_pippenger(pts, sc, 255)) is None
    # device form, 32-byte scalars
    n = 3000
    pts = b"".join(base[i % 40] for i in range(n))
    sc = bytes(rng.getrandbits(8) for _ in range(32 * n))
    cache = m.BatchedBLSVerifierCache.init(max_sets=64)
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    got = m.p2s_mult_pippenger_device(cache, dp.data_ptr(), n, ds.data_ptr(), 255)
    assert o.g2_to_blst_affine(g2_jac_to_affine(got)) == co.msm_g2(pts, sc, 255, 32)
    m.lib().mi355_bls_default_ctx_release()


def test_g2_pippenger_lds_sort_path(m):
    """G2 MSM at n >= 2^15 (the counting sort with the counters in LDS, shared with G1): 64-bit scalars as `combine` passes them
    (core :639-646) and 255-bit ones, against the C restatement."""
    import random
    import c_oracle as co
    from util import g2_jac_to_affine
    rng = random.Random(5)
    h = co.hash_to_g2(b"g2 msm big", o.DST_SIG)
    base = [co.g2_mul(h, rng.randrange(1, o.R)) for _ in range(64)]
    for n, nbits in ((33000, 64), (40000, 255)):
        pts = b"".join(base[i % 64] for i in range(n))
        sb = (nbits + 7) // 8
        sc = bytes(rng.getrandbits(8) for _ in range(sb * n))
        if nbits == 255:
            sc = b"".join(sc[32 * i:32 * i + 31] + bytes([sc[32 * i + 31] & 0x7f]) for i in range(n))
        got = m.blst_p2s_mult_pippenger(pts, sc, nbits)
        assert o.g2_to_blst_affine(g2_jac_to_affine(got)) == co.msm_g2(pts, sc, nbits, sb), (n, nbits)


def test_g2_pippenger_linearity_at_2_18(m):
    """G2 MSM large enough for the two window groups on two streams (n x windows >= 2^22) on top of the LDS counting sort:
    MSM(k) + MSM(k') == MSM(k + k') on the same 2^18 points (scalars below 2^254), checked with the big-int oracle's group law."""
    import random
    import numpy as np
    import c_oracle as co
    from util import g2_jac_to_affine
    rng = random.Random(9)
    n = 1 << 18
    h = co.hash_to_g2(b"g2 msm 2^18", o.DST_SIG)
    base = [co.g2_mul(h, rng.randrange(1, o.R)) for _ in range(256)]
    pts = b"".join(base[i % 256] for i in range(n))
    ra = np.random.default_rng(3)
    k1 = ra.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k2 = ra.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k1[:, 31] &= 0x3f
    k2[:, 31] &= 0x3f
    s = np.zeros((n, 32), dtype=np.uint8)
    carry = np.zeros(n, dtype=np.uint16)
    for j in range(32):
        t = k1[:, j].astype(np.uint16) + k2[:, j].astype(np.uint16) + carry
        s[:, j] = (t & 0xff).astype(np.uint8)
        carry = t >> 8
    a = g2_jac_to_affine(m.blst_p2s_mult_pippenger(pts, k1.tobytes(), 255))
    b = g2_jac_to_affine(m.blst_p2s_mult_pippenger(pts, k2.tobytes(), 255))
    c = g2_jac_to_affine(m.blst_p2s_mult_pippenger(pts, s.tobytes(), 255))
    assert o.g2_add(a, b) == c and c is not None


def test_g2_pippenger_vs_c_oracle_at_2_18(m):
    """G2 MSM at 2^18 points against the C restatement of the bucket method on G2 (oracle_msm_g2_pippenger): 64-bit scalars as
    `combine` passes them (blst_min_pubkey_sig_core.nim:639-646), 8 bytes apart, and 255-bit ones."""
    import random
    import numpy as np
    import c_oracle as co
    from util import g2_jac_to_affine
    rng = random.Random(18)
    n = 1 << 18
    h = co.hash_to_g2(b"g2 msm 2^18 parity", o.DST_SIG)
    base = [co.g2_mul(h, rng.randrange(1, o.R)) for _ in range(256)]
    pts = b"".join(base[i % 256] for i in range(n))
    for nbits, sb in ((64, 8), (255, 32)):
        sc = np.random.default_rng(nbits).integers(0, 256, size=(n, sb), dtype=np.uint8)
        if nbits == 255:
            sc[:, 31] &= 0x7f
        sc = sc.tobytes()
        got = m.blst_p2s_mult_pippenger(pts, sc, nbits)
        assert o.g2_to_blst_affine(g2_jac_to_affine(got)) == co.msm_g2_pippenger(pts, sc, nbits, sb), nbits
