"""ctypes binding of oracle/_build/libbls_oracle.so (the C restatement; checker only)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libbls_oracle.so"))
        vp, sz, i32, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p
        L.oracle_batch_verify.argtypes = [cp, sz, cp, i32, vp, vp, vp, vp, vp]
        L.oracle_hash_to_g2.argtypes = [cp, sz, cp, sz, cp]
        L.oracle_sk_to_pk.argtypes = [cp, cp]
        L.oracle_sign.argtypes = [cp, cp, sz, cp]
        L.oracle_g2_mul.argtypes = [cp, cp, cp]
        L.oracle_make_batch.argtypes = [vp, sz, ctypes.c_uint64]
        L.oracle_make_pks.argtypes = [vp, sz, ctypes.c_uint64]
        L.oracle_deserialize_sets.argtypes = [cp, cp, cp, sz, vp, vp]
        L.oracle_compress_sets.argtypes = [cp, sz, vp, vp, vp]
        L.oracle_deserialize_sets_ex.argtypes = [cp, cp, cp, sz, ctypes.c_uint, vp, vp]
        L.oracle_serialize_sets.argtypes = [cp, sz, vp, vp]
        L.oracle_g1_sum.argtypes = [cp, sz, cp]
        L.oracle_fast_aggregate_verify.argtypes = [cp, sz, cp, sz, cp]
        L.oracle_msm_g1.argtypes = [cp, cp, sz, i32, cp]
        L.oracle_sha256.argtypes = [cp, sz, cp]
        L.oracle_msm_g1_pippenger.argtypes = [cp, cp, sz, i32, cp]
        L.oracle_core_verify.argtypes = [cp, cp, sz, cp]
        L.oracle_msm_g2.argtypes = [cp, cp, sz, i32, i32, cp]
        L.oracle_msm_g2_pippenger.argtypes = [cp, cp, sz, i32, i32, cp]
        L.oracle_aggregate_verify.argtypes = [cp, cp, ctypes.POINTER(ctypes.c_uint32), sz, cp, vp]
        L.oracle_g2_sum.argtypes = [cp, sz, cp]
        L.oracle_combine.argtypes = [cp, cp, cp, sz, cp, cp, vp]
        L.oracle_set_num_threads.argtypes = [i32]
        L.oracle_set_num_threads.restype = None
        _lib = L
    return _lib


def batch_verify(sets, rnd, nthreads, stages=False):
    n = len(sets) // 320
    if not stages:
        return bool(lib().oracle_batch_verify(sets, n, rnd, nthreads, None, None, None, None, None))
    r = (ctypes.c_uint64 * max(n, 1))()
    h = ctypes.create_string_buffer(192 * max(n, 1))
    p = ctypes.create_string_buffer(96 * max(n, 1))
    a = ctypes.create_string_buffer(192)
    g = ctypes.create_string_buffer(576)
    ok = bool(lib().oracle_batch_verify(sets, n, rnd, nthreads, r, h, p, a, g))
    return ok, {"r": list(r)[:n], "H": h.raw, "rPK": p.raw, "aggsig": a.raw, "gt": g.raw}


def make_batch(n, seed=0):
    b = ctypes.create_string_buffer(320 * n)
    lib().oracle_make_batch(b, n, seed)
    return b.raw


def hash_to_g2(msg, dst):
    o = ctypes.create_string_buffer(192)
    lib().oracle_hash_to_g2(msg, len(msg), dst, len(dst), o)
    return o.raw


def sk_to_pk(sk_int):
    o = ctypes.create_string_buffer(96)
    lib().oracle_sk_to_pk(sk_int.to_bytes(32, "little"), o)
    return o.raw


def sign(sk_int, msg):
    o = ctypes.create_string_buffer(192)
    lib().oracle_sign(sk_int.to_bytes(32, "little"), msg, len(msg), o)
    return o.raw


def g2_mul(p192, k_int):
    o = ctypes.create_string_buffer(192)
    lib().oracle_g2_mul(p192, k_int.to_bytes(32, "little"), o)
    return o.raw


def g1_sum(pts):
    o = ctypes.create_string_buffer(96)
    lib().oracle_g1_sum(pts, len(pts) // 96, o)
    return o.raw


def fast_aggregate_verify(pks, msg, sig):
    return bool(lib().oracle_fast_aggregate_verify(pks, len(pks) // 96, msg, len(msg), sig))


def msm_g1(pts, scalars, nbits=255):
    o = ctypes.create_string_buffer(96)
    lib().oracle_msm_g1(pts, scalars, len(pts) // 96, nbits, o)
    return o.raw


def msm_g1_pippenger(pts, scalars, nbits=255):
    o = ctypes.create_string_buffer(96)
    lib().oracle_msm_g1_pippenger(pts, scalars, len(pts) // 96, nbits, o)
    return o.raw


def core_verify(pk, msg, sig):
    return bool(lib().oracle_core_verify(pk, msg, len(msg), sig))


def set_num_threads(n):
    lib().oracle_set_num_threads(n)


def make_pks(n, seed=0):
    """(pks, sum of the secret keys mod r) for the keys oracle_make_batch derives."""
    import hashlib
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    b = ctypes.create_string_buffer(96 * n)
    lib().oracle_make_pks(b, n, seed)
    tot = 0
    for i in range(n):
        sk = bytearray(hashlib.sha256(b"sk" + (seed + i).to_bytes(8, "little")).digest())
        sk[31] &= 0x3f
        sk[0] |= 1
        tot += int.from_bytes(sk, "little")
    return b.raw, tot % R


def compress_sets(sets):
    n = len(sets) // 320
    pk, ms, sg = ctypes.create_string_buffer(48 * n), ctypes.create_string_buffer(32 * n), ctypes.create_string_buffer(96 * n)
    lib().oracle_compress_sets(sets, n, pk, ms, sg)
    return pk.raw, ms.raw, sg.raw


def deserialize_sets(pks, msgs, sigs):
    n = len(pks) // 48
    out, st = ctypes.create_string_buffer(320 * n), ctypes.create_string_buffer(n)
    ok = lib().oracle_deserialize_sets(pks, msgs, sigs, n, out, st)
    return bool(ok), out.raw, st.raw


def serialize_sets(sets):
    """Uncompressed wire form of the records: (n x 96-byte keys, n x 192-byte signatures)."""
    n = len(sets) // 320
    pk, sg = ctypes.create_string_buffer(96 * n), ctypes.create_string_buffer(192 * n)
    lib().oracle_serialize_sets(sets, n, pk, sg)
    return pk.raw, sg.raw


def deserialize_sets_ex(pks, msgs, sigs, flags):
    n = len(msgs) // 32
    out, st = ctypes.create_string_buffer(320 * n), ctypes.create_string_buffer(n)
    ok = lib().oracle_deserialize_sets_ex(pks, msgs, sigs, n, flags, out, st)
    return bool(ok), out.raw, st.raw


def msm_g2(pts, scalars, nbits=255, sbytes=32):
    o = ctypes.create_string_buffer(192)
    lib().oracle_msm_g2(pts, scalars, len(pts) // 192, sbytes, nbits, o)
    return o.raw


def msm_g2_pippenger(pts, scalars, nbits=255, sbytes=32):
    o = ctypes.create_string_buffer(192)
    lib().oracle_msm_g2_pippenger(pts, scalars, len(pts) // 192, sbytes, nbits, o)
    return o.raw


def aggregate_verify(pks, msgs, sig, gt=False):
    """pks: list of 96-byte keys (or their concatenation), msgs: list of byte strings, sig: 192 bytes -> verdict (and the GT value)."""
    pkb = pks if isinstance(pks, (bytes, bytearray)) else b"".join(pks)
    offs = [0]
    for x in msgs:
        offs.append(offs[-1] + len(x))
    arr = (ctypes.c_uint32 * len(offs))(*offs)
    g = ctypes.create_string_buffer(576)
    ok = bool(lib().oracle_aggregate_verify(bytes(pkb), b"".join(msgs) or b"\0", arr, len(msgs), sig, g))
    return (ok, g.raw) if gt else ok


def g2_sum(pts):
    o = ctypes.create_string_buffer(192)
    lib().oracle_g2_sum(pts, len(pts) // 192, o)
    return o.raw


def combine(rnd, pks, sigs):
    """-> (out_pk96, out_sig192, [scalars])"""
    n = len(pks) // 96
    pk, sg = ctypes.create_string_buffer(96), ctypes.create_string_buffer(192)
    sc = (ctypes.c_uint64 * n)()
    lib().oracle_combine(rnd, pks, sigs, n, pk, sg, sc)
    return pk.raw, sg.raw, list(sc)
