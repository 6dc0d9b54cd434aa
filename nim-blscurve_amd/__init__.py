"""Host-side mirror of nim-blscurve's batch-verifier API over the MI355X C ABI.

Python stands in for the Nim host layer (no Nim toolchain in the build image); names, argument
meaning and error behaviour follow ``blscurve/bls_batch_verifier.nim``:

  SignatureSet            (pubkey, message[32], signature) triplet, 320-byte record  (:34)
  BatchedBLSVerifierCache reusable per-caller scratch -> persistent device workspace  (:62-69,:108-119)
  batchVerifySerial       (:121-177)      batchVerifyParallel (:296-416)      batchVerify (:420-495)

The compute lives entirely in ``libblscurve_mi355x.so`` (hand-written HIP, gfx950).  There is no
CPU fallback: importing works without a GPU (so symbols can be checked), every compute call
raises ``BlsGpuError`` if the library or a GPU is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_BLS_LIB: another build of the same library (the A/B scripts under tools/ point it at nim-blscurve_amd/variants/<name>.so
# instead of copying a variant over the shipped file)
LIB_PATH = os.environ.get("MI355_BLS_LIB") or os.path.join(_HERE, "libblscurve_mi355x.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "blscurve_mi355x.h"))

SIGSET_BYTES = 320
DEFAULT_NUM_THREADS = 4096


class BlsGpuError(RuntimeError):
    pass


_lib = None


def lib():
    """The C-ABI library; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BlsGpuError("HIP extension missing: %s (run nim-blscurve_amd/build.sh); no CPU fallback exists" % LIB_PATH)
        # torch bundles its own HIP runtime with the same soname as /opt/rocm's: whichever loads first
        # serves both, and torch must be that one or it later reports "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
        L.mi355_bls_ctx_create.argtypes = [ctypes.POINTER(vp), i32, sz]
        L.mi355_bls_ctx_destroy.argtypes = [vp]
        L.mi355_bls_ctx_destroy.restype = None
        L.mi355_bls_last_error.restype = ctypes.c_char_p
        L.mi355_bls_build_info.restype = ctypes.c_char_p
        L.mi355_bls_ctx_set_num_threads.argtypes = [vp, u32]
        L.mi355_bls_ctx_set_cooperative.argtypes = [vp, i32]
        L.mi355_bls_batch_verify.argtypes = [vp, vp, sz, ctypes.c_char_p]
        L.mi355_bls_batch_verify_serial.argtypes = [vp, vp, sz, ctypes.c_char_p]
        L.mi355_bls_batch_verify_device.argtypes = [vp, vp, sz, ctypes.c_char_p, vp]
        L.mi355_bls_batch_submit_device.argtypes = [vp, vp, sz, ctypes.c_char_p, vp, vp]
        L.mi355_bls_batch_wait.argtypes = [vp]
        L.mi355_bls_batch_verify_many.argtypes = [vp, vp, ctypes.POINTER(sz), ctypes.c_char_p, sz, ctypes.c_char_p]
        L.mi355_bls_batch_verify_many_device.argtypes = [vp, vp, ctypes.POINTER(sz), ctypes.c_char_p, sz, ctypes.c_char_p, vp]
        L.mi355_bls_batch_shard_device.argtypes = [vp, vp, sz, u32, u32, ctypes.c_char_p, vp, ctypes.c_char_p, ctypes.POINTER(i32)]
        L.mi355_bls_batch_shard_submit_device.argtypes = [vp, vp, sz, u32, u32, ctypes.c_char_p, vp, vp]
        L.mi355_bls_batch_shard_wait.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i32)]
        L.mi355_bls_finalverify_shards.argtypes = [vp, ctypes.c_char_p, sz]
        L.mi355_bls_chunk_range.argtypes = [sz, u32, u32, u32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.mi355_bls_chunk_range.restype = None
        L.mi355_bls_ctx_shard_blob_device.argtypes = [vp, ctypes.POINTER(vp)]
        L.mi355_bls_ctx_set_shard_blob_device.argtypes = [vp, vp]
        L.mi355_bls_finalverify_blobs_submit_device.argtypes = [vp, vp, sz, sz, vp]
        L.mi355_bls_finalverify_wait.argtypes = [vp]
        L.mi355_bls_shard_plan.argtypes = [sz, u32, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.mi355_bls_batch_verify_multi.argtypes = [ctypes.POINTER(vp), sz, vp, sz, ctypes.c_char_p]
        L.mi355_bls_batch_verify_multi_device.argtypes = [ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz, ctypes.c_char_p]
        L.mi355_bls_batch_verify_once.argtypes = [vp, sz, ctypes.c_char_p, u32]
        L.mi355_bls_default_ctx_release.argtypes = []
        L.mi355_bls_default_ctx_release.restype = None
        L.mi355_p1s_mult_pippenger_scratch_sizeof.argtypes = [sz]
        L.mi355_p1s_mult_pippenger_scratch_sizeof.restype = sz
        L.mi355_p1s_mult_pippenger.argtypes = [vp, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz, vp]
        L.mi355_p1s_mult_pippenger.restype = None
        L.mi355_p2s_mult_pippenger_scratch_sizeof.argtypes = [sz]
        L.mi355_p2s_mult_pippenger_scratch_sizeof.restype = sz
        L.mi355_p2s_mult_pippenger.argtypes = [vp, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz, vp]
        L.mi355_p2s_mult_pippenger.restype = None
        L.mi355_bls_p2s_mult_pippenger.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz]
        L.mi355_bls_p2s_mult_pippenger_device.argtypes = [vp, ctypes.c_char_p, vp, sz, vp, sz, vp]
        L.mi355_bls_g1_aggregate.argtypes = [vp, vp, sz, ctypes.c_char_p]
        L.mi355_bls_g1_aggregate_device.argtypes = [vp, vp, sz, vp, ctypes.c_char_p]
        L.mi355_bls_g2_aggregate.argtypes = [vp, vp, sz, ctypes.c_char_p]
        L.mi355_bls_g2_aggregate_device.argtypes = [vp, vp, sz, vp, ctypes.c_char_p]
        L.mi355_bls_recommend_hw_queues.argtypes = []
        L.mi355_bls_fast_aggregate_verify.argtypes = [vp, vp, sz, ctypes.c_char_p, sz, ctypes.c_char_p]
        L.mi355_bls_fast_aggregate_verify_device.argtypes = [vp, vp, sz, ctypes.c_char_p, sz, ctypes.c_char_p, vp]
        L.mi355_bls_fast_aggregate_verify_multi.argtypes = [ctypes.POINTER(vp), sz, vp, sz, ctypes.c_char_p, sz, ctypes.c_char_p]
        L.mi355_bls_verify_aggregate.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, sz, ctypes.c_char_p]
        L.mi355_bls_p1s_mult_pippenger_scratch_sizeof.argtypes = [sz]
        L.mi355_bls_p1s_mult_pippenger_scratch_sizeof.restype = sz
        L.mi355_bls_p1s_mult_pippenger.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz]
        L.mi355_bls_p1s_mult_pippenger_device.argtypes = [vp, ctypes.c_char_p, vp, sz, vp, sz, vp]
        cp = ctypes.c_char_p
        L.mi355_bls_deserialize_sets.argtypes = [vp, cp, cp, cp, sz, vp, vp]
        L.mi355_bls_deserialize_sets_device.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
        L.mi355_bls_deserialize_sets_ex.argtypes = [vp, cp, cp, cp, sz, u32, vp, vp]
        L.mi355_bls_deserialize_sets_ex_device.argtypes = [vp, vp, vp, vp, sz, u32, vp, vp, vp]
        L.mi355_bls_batch_verify_compressed.argtypes = [vp, cp, cp, cp, sz, cp, vp]
        L.mi355_bls_batch_verify_compressed_device.argtypes = [vp, vp, vp, vp, sz, cp, vp, vp]
        L.mi355_bls_last_deser_ms.argtypes = [vp]
        L.mi355_bls_last_deser_ms.restype = ctypes.c_float
        L.mi355_bls_combine.argtypes = [vp, cp, cp, cp, sz, cp, cp]
        L.mi355_bls_sign_sets.argtypes = [vp, cp, cp, sz, vp, vp]
        L.mi355_bls_sign_sets_device.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.mi355_bls_aggregate_verify.argtypes = [vp, cp, cp, ctypes.POINTER(ctypes.c_uint32), sz, cp]
        L.mi355_bls_aggv_init.argtypes = [vp]
        L.mi355_bls_aggv_update.argtypes = [vp, cp, cp, sz]
        L.mi355_bls_aggv_finish.argtypes = [vp, cp]
        L.mi355_bls_aggv_finish_p2.argtypes = [vp, cp]
        L.mi355_bls_aggregate_verify_p2.argtypes = [vp, cp, cp, ctypes.POINTER(ctypes.c_uint32), sz, cp]
        L.mi355_bls_msm_shard_range.argtypes = [sz, u32, u32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.mi355_bls_msm_shard_range.restype = None
        L.mi355_bls_p1s_mult_pippenger_multi.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_char_p, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz]
        L.mi355_bls_p2s_mult_pippenger_multi.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_char_p, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz]
        L.mi355_bls_p1s_mult_pippenger_multi_device.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_char_p, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz]
        L.mi355_bls_p1s_mult_pippenger_partial_device.argtypes = [vp, vp, vp, sz, vp, sz, vp]
        L.mi355_bls_p1s_add.argtypes = [vp, ctypes.c_char_p, cp, sz]
        L.mi355_bls_p2s_add.argtypes = [vp, ctypes.c_char_p, cp, sz]
        L.mi355_bls_p1s_add_device.argtypes = [vp, ctypes.c_char_p, vp, sz, sz, vp]
        L.mi355_bls_debug_fail_next_enqueue.argtypes = [vp]
        L.mi355_bls_debug_batches_in_flight.argtypes = []
        L.mi355_bls_last_fold_form.argtypes = [vp]
        L.mi355_bls_debug_g2_clear_cofactor.argtypes = [vp, cp, sz, cp]
        L.mi355_bls_debug_hash_to_g2.argtypes = [vp, cp, sz, cp, sz, cp]
        L.mi355_bls_debug_multi_enqueue_us.argtypes = [ctypes.POINTER(ctypes.c_float), sz]
        L.mi355_bls_debug_multi_enqueue_us.restype = sz
        L.mi355_bls_fetch_stage.argtypes = [vp, i32, vp, sz]
        L.mi355_bls_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.mi355_bls_last_kernel_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        _lib = L
    return _lib


def build_info():
    """mi355_bls_build_info() as a dict: {"aligned": bool, "dpp_combine_off": bool, "stamp": str}."""
    kv = dict(x.split("=", 1) for x in lib().mi355_bls_build_info().decode().split())
    return {"aligned": kv.get("aligned") == "1", "dpp_combine_off": kv.get("dpp_combine") == "off", "stamp": kv.get("stamp", "unknown")}


def _check(rc):
    if rc < 0:
        raise BlsGpuError("mi355_bls error %d: %s" % (rc, lib().mi355_bls_last_error().decode()))
    return rc


def _rnd32(secureRandomBytes):
    """secureRandomBytes is `array[32, byte]` in the reference (bls_batch_verifier.nim:301): exactly 32 bytes,
    bytes-like.  (bytes(int) would silently give that many ZERO bytes, a short buffer lets the C side read past it.)"""
    if not isinstance(secureRandomBytes, (bytes, bytearray, memoryview)):
        raise ValueError("secureRandomBytes must be a bytes-like object of 32 bytes")
    b = bytes(secureRandomBytes)
    if len(b) != 32:
        raise ValueError("secureRandomBytes must be exactly 32 bytes, got %d" % len(b))
    return b


def pack_signature_sets(sets):
    """[(pubkey96, message32, signature192)] -> contiguous 320-byte records (the Nim tuple layout)."""
    out = bytearray()
    for pk, msg, sig in sets:
        if len(pk) != 96 or len(msg) != 32 or len(sig) != 192:
            raise ValueError("SignatureSet = (96-byte blst_p1_affine, 32-byte message, 192-byte blst_p2_affine)")
        out += pk + msg + sig
    return bytes(out)


def _as_records(input_):
    if isinstance(input_, (bytes, bytearray, memoryview)):
        b = bytes(input_)
        if len(b) % SIGSET_BYTES:
            raise ValueError("record buffer is not a multiple of 320 bytes")
        return b
    return pack_signature_sets(input_)


def chunk_range(n_total, num_threads, chunk_lo, chunk_hi):
    """Tuple range of chunks [chunk_lo, chunk_hi) (parallel_chunks.nim:42-66)."""
    first, count = ctypes.c_size_t(), ctypes.c_size_t()
    lib().mi355_bls_chunk_range(n_total, num_threads, chunk_lo, chunk_hi, ctypes.byref(first), ctypes.byref(count))
    return first.value, count.value


class BatchedBLSVerifierCache:
    """bls_batch_verifier.nim:62-69.  ``init(numThreads=...)`` mirrors ``init(tp: Taskpool)``:
    numThreads is the number of blinding chains the parallel path uses (B = min(n, numThreads))."""

    def __init__(self, max_sets=65536, numThreads=DEFAULT_NUM_THREADS, device=0):
        self._h = ctypes.c_void_p()
        self.max_sets = max_sets
        self.numThreads = numThreads
        _check(lib().mi355_bls_ctx_create(ctypes.byref(self._h), device, max_sets))
        _check(lib().mi355_bls_ctx_set_num_threads(self._h, numThreads))

    @classmethod
    def init(cls, max_sets=65536, numThreads=DEFAULT_NUM_THREADS, device=0):
        return cls(max_sets, numThreads, device)

    def close(self):
        if self._h:
            lib().mi355_bls_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stage outputs of the last call (parity tests) --
    def set_cooperative(self, on):
        """Small batches: 8 lanes per set (latency, default) or one lane per set (throughput with many batches in flight)."""
        _check(lib().mi355_bls_ctx_set_cooperative(self._h, 1 if on else 0))

    def fetch(self, what, nbytes):
        b = ctypes.create_string_buffer(nbytes)
        _check(lib().mi355_bls_fetch_stage(self._h, what, b, nbytes))
        return b.raw

    def timings(self):
        t = (ctypes.c_float * 8)()
        _check(lib().mi355_bls_last_timings(self._h, t))
        names = ["blinding", "hash_to_g2", "pk_mul", "sig_mul_sum", "miller_lines", "line_products", "final", "total"]
        return dict(zip(names, list(t)))

    def kernel_timings(self):
        """ms of the kernels inside the two-kernel stages of the last batch call."""
        t = (ctypes.c_float * 4)()
        _check(lib().mi355_bls_last_kernel_timings(self._h, t))
        return dict(zip(["k_hash_map", "k_hash_clear", "k_lineprod", "k_lineprod2"], list(t)))

    def fold_form(self):
        """which fold of the line products the last batch call enqueued: 1 = the Fp12 engine (k_fold), 0 = k_lineprod2"""
        return lib().mi355_bls_last_fold_form(self._h)

    # -- device-resident entry points --
    def verify_device(self, d_ptr, n, secureRandomBytes, stream=0):
        return bool(_check(lib().mi355_bls_batch_verify_device(self._h, d_ptr, n, _rnd32(secureRandomBytes), stream)))

    def submit_device(self, d_ptr, n, secureRandomBytes, stream=0, after=None):
        """Enqueue a batch verification and return at once; wait() gives its verdict.  after: a context whose
        batch was submitted before; this one then starts beside that batch's serial tail."""
        _check(lib().mi355_bls_batch_submit_device(self._h, d_ptr, n, _rnd32(secureRandomBytes), stream, after._h if after is not None else None))

    def wait(self):
        return bool(_check(lib().mi355_bls_batch_wait(self._h)))

    def shard_device(self, d_ptr, n_total, chunk_lo, chunk_hi, secureRandomBytes, stream=0):
        out = ctypes.create_string_buffer(576)
        ok = ctypes.c_int()
        _check(lib().mi355_bls_batch_shard_device(self._h, d_ptr, n_total, chunk_lo, chunk_hi, _rnd32(secureRandomBytes), stream, out, ctypes.byref(ok)))
        return out.raw, bool(ok.value)

    def shard_submit_device(self, d_ptr, n_total, chunk_lo, chunk_hi, secureRandomBytes, stream=0, after=None):
        _check(lib().mi355_bls_batch_shard_submit_device(self._h, d_ptr, n_total, chunk_lo, chunk_hi, _rnd32(secureRandomBytes), stream,
                                                         after._h if after is not None else None))

    def shard_wait(self):
        out = ctypes.create_string_buffer(576)
        ok = ctypes.c_int()
        _check(lib().mi355_bls_batch_shard_wait(self._h, out, ctypes.byref(ok)))
        return out.raw, bool(ok.value)

    BLOB_BYTES = 640

    def shard_blob_ptr(self):
        """Device address of this context's shard blob (state | ok word), written by every shard submit on its stream."""
        p = ctypes.c_void_p()
        _check(lib().mi355_bls_ctx_shard_blob_device(self._h, ctypes.byref(p)))
        return p.value

    def set_shard_blob_ptr(self, d_ptr):
        """Shard submits write their blob (640 bytes) to this device address instead (the collective's send buffer); None resets."""
        _check(lib().mi355_bls_ctx_set_shard_blob_device(self._h, d_ptr))

    def finalverify_blobs_submit(self, d_blobs, k, stride=640, stream=0):
        """merge + finalVerify on k gathered shard blobs resident in device memory; finalverify_wait() gives the verdict."""
        _check(lib().mi355_bls_finalverify_blobs_submit_device(self._h, d_blobs, k, stride, stream))

    def finalverify_wait(self):
        return bool(_check(lib().mi355_bls_finalverify_wait(self._h)))

    def finalverify_shards(self, states):
        blob = b"".join(states)
        return bool(_check(lib().mi355_bls_finalverify_shards(self._h, blob, len(states))))


def shard_plan(n_total, num_threads, world, rank):
    """(chunk_lo, chunk_hi, first_tuple, tuple_count) of device `rank` (mi355_bls_shard_plan)."""
    lo, hi, first, count = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_size_t(), ctypes.c_size_t()
    _check(lib().mi355_bls_shard_plan(n_total, num_threads, world, rank, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(first), ctypes.byref(count)))
    return lo.value, hi.value, first.value, count.value


def batchVerifyMulti(caches, input_, secureRandomBytes):
    """batchVerifyParallel over several devices from one host thread (mi355_bls_batch_verify_multi): caches[g] lives on
    device g; all must share numThreads.  Empty input -> False."""
    rec = _as_records(input_)
    n = len(rec) // SIGSET_BYTES
    if n == 0:
        return False
    arr = (ctypes.c_void_p * len(caches))(*[c._h for c in caches])
    return bool(_check(lib().mi355_bls_batch_verify_multi(arr, len(caches), rec, n, _rnd32(secureRandomBytes))))


def batchVerifyMulti_device(caches, d_ptrs, n, secureRandomBytes):
    """Same with shard g's records already resident on device g (d_ptrs[g])."""
    arr = (ctypes.c_void_p * len(caches))(*[c._h for c in caches])
    ptrs = (ctypes.c_void_p * len(caches))(*d_ptrs)
    return bool(_check(lib().mi355_bls_batch_verify_multi_device(arr, len(caches), ptrs, n, _rnd32(secureRandomBytes))))


def batchVerifyOnce(input_, secureRandomBytes, numThreads=DEFAULT_NUM_THREADS):
    """The cache-less overloads batchVerify(tp, input, rnd) (bls_batch_verifier.nim:475-495) on the process-wide default context."""
    rec = _as_records(input_)
    n = len(rec) // SIGSET_BYTES
    if n == 0:
        return False
    return bool(_check(lib().mi355_bls_batch_verify_once(rec, n, _rnd32(secureRandomBytes), numThreads)))


def batchVerifyMany(cache, inputs, secureRandomBytes_list):
    """k independent batches in ONE device pass (mi355_bls_batch_verify_many): inputs[b] is batch b (records or SignatureSet list),
    secureRandomBytes_list[b] its random bytes.  -> [bool] * k, each what batchVerify(cache, inputs[b], rnd[b]) returns."""
    recs = [_as_records(x) for x in inputs]
    if len(recs) != len(secureRandomBytes_list):
        raise ValueError("one secureRandomBytes per batch")
    k = len(recs)
    if k == 0:
        return []
    counts = (ctypes.c_size_t * k)(*[len(r) // SIGSET_BYTES for r in recs])
    rnds = b"".join(_rnd32(r) for r in secureRandomBytes_list)
    out = ctypes.create_string_buffer(k)
    _check(lib().mi355_bls_batch_verify_many(cache._h, b"".join(recs) or b"\0", counts, rnds, k, out))
    return [bool(v) for v in out.raw]


def batchVerifyMany_device(cache, d_ptr, counts, secureRandomBytes_list, stream=0):
    """Same with the tuples of all batches contiguous in device memory."""
    k = len(counts)
    carr = (ctypes.c_size_t * k)(*counts)
    rnds = b"".join(_rnd32(r) for r in secureRandomBytes_list)
    out = ctypes.create_string_buffer(k)
    _check(lib().mi355_bls_batch_verify_many_device(cache._h, d_ptr, carr, rnds, k, out, stream))
    return [bool(v) for v in out.raw]


def batchVerifySerial(cache, input_, secureRandomBytes):
    """bls_batch_verifier.nim:121-160.  Empty input -> False."""
    rec = _as_records(input_)
    n = len(rec) // SIGSET_BYTES
    if n == 0:
        return False
    return bool(_check(lib().mi355_bls_batch_verify_serial(cache._h, rec, n, _rnd32(secureRandomBytes))))


def batchVerifyParallel(cache, input_, secureRandomBytes):
    """bls_batch_verifier.nim:296-416."""
    rec = _as_records(input_)
    n = len(rec) // SIGSET_BYTES
    if n == 0:
        return False
    return bool(_check(lib().mi355_bls_batch_verify(cache._h, rec, n, _rnd32(secureRandomBytes))))


def batchVerify(cache, input_, secureRandomBytes):
    """bls_batch_verifier.nim:420-495: parallel iff numThreads > 1 and n >= 3, else serial."""
    rec = _as_records(input_)
    n = len(rec) // SIGSET_BYTES
    if cache.numThreads > 1 and n >= 3:
        return batchVerifyParallel(cache, rec, secureRandomBytes)
    return batchVerifySerial(cache, rec, secureRandomBytes)


def aggregateAll(cache, publicKeys):
    """G1 aggregateAll (blst_min_pubkey_sig_core.nim:179-195): n x 96-byte affine keys -> 144-byte blst_p1.
    Empty input -> None (the reference returns false)."""
    buf = bytes(publicKeys) if isinstance(publicKeys, (bytes, bytearray, memoryview)) else b"".join(publicKeys)
    if len(buf) % 96:
        raise ValueError("public keys are 96-byte blst_p1_affine images")
    n = len(buf) // 96
    if n == 0:
        return None
    out = ctypes.create_string_buffer(144)
    _check(lib().mi355_bls_g1_aggregate(cache._h, buf, n, out))
    return out.raw


def aggregateAllSignatures(cache, signatures):
    """aggregateAll on signatures (genAggregatorProcedures(AggregateSignature, Signature, p2), blst_min_pubkey_sig_core.nim:179-195,211):
    n x 192-byte affine signatures -> 288-byte blst_p2 (an AggregateSignature).  Empty input -> None (the reference returns false)."""
    buf = bytes(signatures) if isinstance(signatures, (bytes, bytearray, memoryview)) else b"".join(signatures)
    if len(buf) % 192:
        raise ValueError("signatures are 192-byte blst_p2_affine images")
    n = len(buf) // 192
    if n == 0:
        return None
    out = ctypes.create_string_buffer(288)
    _check(lib().mi355_bls_g2_aggregate(cache._h, buf, n, out))
    return out.raw


def fastAggregateVerify(cache, publicKeys, message, signature):
    """bls_sig_min_pubkey.nim:234-258.  Empty key list -> False."""
    buf = bytes(publicKeys) if isinstance(publicKeys, (bytes, bytearray, memoryview)) else b"".join(publicKeys)
    if len(buf) % 96 or len(signature) != 192:
        raise ValueError("public keys are 96-byte, the signature a 192-byte BLST affine image")
    n = len(buf) // 96
    if n == 0:
        return False
    return bool(_check(lib().mi355_bls_fast_aggregate_verify(cache._h, buf, n, bytes(message), len(message), bytes(signature))))


def fastAggregateVerifyMulti(caches, publicKeys, message, signature):
    """fastAggregateVerify with the keys sharded over several devices (caches[g] on device g), one pairing on caches[0]."""
    buf = bytes(publicKeys) if isinstance(publicKeys, (bytes, bytearray, memoryview)) else b"".join(publicKeys)
    if len(buf) % 96 or len(signature) != 192:
        raise ValueError("public keys are 96-byte, the signature a 192-byte BLST affine image")
    n = len(buf) // 96
    if n == 0:
        return False
    arr = (ctypes.c_void_p * len(caches))(*[c._h for c in caches])
    return bool(_check(lib().mi355_bls_fast_aggregate_verify_multi(arr, len(caches), buf, n, bytes(message), len(message), bytes(signature))))


def verifyAggregate(cache, aggregate_p1, message, signature):
    """coreVerifyNoGroupCheck (core :269-297) on a 144-byte blst_p1 aggregate public key the caller holds."""
    if len(aggregate_p1) != 144 or len(signature) != 192:
        raise ValueError("aggregate: 144-byte blst_p1, signature: 192-byte blst_p2_affine")
    return bool(_check(lib().mi355_bls_verify_aggregate(cache._h, bytes(aggregate_p1), bytes(message), len(message), bytes(signature))))


def p1s_mult_pippenger(cache, points, scalars, nbits=255):
    """blst_p1s_mult_pippenger shape (benchmarks/bls12381_msm_g1.nim:50-59): points = n x 96-byte affine,
    scalars = n x 32-byte little-endian; returns the 144-byte blst_p1 (Jacobian) result."""
    if len(points) % 96 or len(scalars) % 32 or len(points) // 96 != len(scalars) // 32:
        raise ValueError("points: n x 96 bytes, scalars: n x 32 bytes")
    n = len(points) // 96
    pb = ctypes.create_string_buffer(bytes(points), len(points)) if n else None
    sb = ctypes.create_string_buffer(bytes(scalars), len(scalars)) if n else None
    pl = (ctypes.c_void_p * 2)(ctypes.addressof(pb) if n else None, None)       # [ptr, NULL]
    sl = (ctypes.c_void_p * 2)(ctypes.addressof(sb) if n else None, None)
    out = ctypes.create_string_buffer(144)
    _check(lib().mi355_bls_p1s_mult_pippenger(cache._h, out, pl, n, sl, nbits))
    return out.raw


def blst_p1s_mult_pippenger(points, scalars, nbits=255, per_element_pointers=False, g2=False):
    """mi355_p1s_mult_pippenger / mi355_p2s_mult_pippenger (g2=True): EXACTLY blst_pNs_mult_pippenger's argument list (no
    context, void, scalars (nbits + 7) // 8 bytes apart, NULL-terminated pointer lists).  per_element_pointers: pass one
    pointer per element instead of [ptr, NULL] (both are blst conventions).  Returns the 144-byte blst_p1 / 288-byte blst_p2."""
    sb = (nbits + 7) // 8
    ab = 192 if g2 else 96
    fn = lib().mi355_p2s_mult_pippenger if g2 else lib().mi355_p1s_mult_pippenger
    if len(points) % ab or len(scalars) % sb or len(points) // ab != len(scalars) // sb:
        raise ValueError("points: n x %d bytes, scalars: n x %d bytes" % (ab, sb))
    n = len(points) // ab
    out = ctypes.create_string_buffer(288 if g2 else 144)
    if n == 0:
        fn(out, None, 0, None, nbits, None)
        return out.raw
    pb = ctypes.create_string_buffer(bytes(points), len(points))
    scb = ctypes.create_string_buffer(bytes(scalars), len(scalars))
    pa, sa = ctypes.addressof(pb), ctypes.addressof(scb)
    if per_element_pointers:
        pl = (ctypes.c_void_p * n)(*[pa + ab * i for i in range(n)])
        sl = (ctypes.c_void_p * n)(*[sa + sb * i for i in range(n)])
    else:
        pl = (ctypes.c_void_p * 2)(pa, None)
        sl = (ctypes.c_void_p * 2)(sa, None)
    scratch = ctypes.create_string_buffer(max(8, lib().mi355_p1s_mult_pippenger_scratch_sizeof(n)))
    fn(out, pl, n, sl, nbits, scratch)
    return out.raw


def blst_p2s_mult_pippenger(points, scalars, nbits=255, per_element_pointers=False):
    return blst_p1s_mult_pippenger(points, scalars, nbits, per_element_pointers, g2=True)


def p2s_mult_pippenger_device(cache, d_points, n, d_scalars, nbits=255, stream=0):
    out = ctypes.create_string_buffer(288)
    _check(lib().mi355_bls_p2s_mult_pippenger_device(cache._h, out, d_points, n, d_scalars, nbits, stream))
    return out.raw


def p1s_mult_pippenger_device(cache, d_points, n, d_scalars, nbits=255, stream=0):
    out = ctypes.create_string_buffer(144)
    _check(lib().mi355_bls_p1s_mult_pippenger_device(cache._h, out, d_points, n, d_scalars, nbits, stream))
    return out.raw


def _split_compressed(pubkeys, messages, signatures):
    pk = bytes(pubkeys) if isinstance(pubkeys, (bytes, bytearray, memoryview)) else b"".join(pubkeys)
    ms = bytes(messages) if isinstance(messages, (bytes, bytearray, memoryview)) else b"".join(messages)
    sg = bytes(signatures) if isinstance(signatures, (bytes, bytearray, memoryview)) else b"".join(signatures)
    if len(pk) % 48 or len(ms) % 32 or len(sg) % 96 or not (len(pk) // 48 == len(ms) // 32 == len(sg) // 96):
        raise ValueError("n x 48-byte compressed keys, n x 32-byte messages, n x 96-byte compressed signatures")
    return pk, ms, sg, len(pk) // 48


def deserializeSets(cache, pubkeys, messages, signatures):
    """Batched PublicKey.fromBytes / Signature.fromBytes (bls_sig_io.nim:42-58,81-99).
    -> (all_ok, n x 320-byte SignatureSet records, per-tuple status bytes)."""
    pk, ms, sg, n = _split_compressed(pubkeys, messages, signatures)
    if n == 0:
        return True, b"", b""
    out = ctypes.create_string_buffer(320 * n)
    st = ctypes.create_string_buffer(n)
    ok = _check(lib().mi355_bls_deserialize_sets(cache._h, pk, ms, sg, n, out, st))
    return bool(ok), out.raw, st.raw


DESER_PK_UNCOMPRESSED, DESER_SIG_UNCOMPRESSED, DESER_KNOWN_ON_CURVE = 1, 2, 4


def deserializeSetsEx(cache, pubkeys, messages, signatures, pk_uncompressed=False, sig_uncompressed=False, known_on_curve=False):
    """The other forms of fromBytes (bls_sig_io.nim:42-121): 96-byte keys / 192-byte signatures (blst_pN_deserialize) and
    fromBytesKnownOnCurve (no subgroup checks).  -> (all_ok, records, status bytes)."""
    pk = bytes(pubkeys) if isinstance(pubkeys, (bytes, bytearray, memoryview)) else b"".join(pubkeys)
    ms = bytes(messages) if isinstance(messages, (bytes, bytearray, memoryview)) else b"".join(messages)
    sg = bytes(signatures) if isinstance(signatures, (bytes, bytearray, memoryview)) else b"".join(signatures)
    pkb, sgb = (96 if pk_uncompressed else 48), (192 if sig_uncompressed else 96)
    n = len(ms) // 32
    if len(ms) % 32 or len(pk) != pkb * n or len(sg) != sgb * n:
        raise ValueError("n x %d-byte keys, n x 32-byte messages, n x %d-byte signatures" % (pkb, sgb))
    if n == 0:
        return True, b"", b""
    flags = (DESER_PK_UNCOMPRESSED if pk_uncompressed else 0) | (DESER_SIG_UNCOMPRESSED if sig_uncompressed else 0) | (DESER_KNOWN_ON_CURVE if known_on_curve else 0)
    out = ctypes.create_string_buffer(320 * n)
    st = ctypes.create_string_buffer(n)
    ok = _check(lib().mi355_bls_deserialize_sets_ex(cache._h, pk, ms, sg, n, flags, out, st))
    return bool(ok), out.raw, st.raw


def batchVerifyCompressed(cache, pubkeys, messages, signatures, secureRandomBytes):
    """fromBytes for every tuple, then batchVerify, on the device.  -> (verdict, per-tuple status bytes)."""
    pk, ms, sg, n = _split_compressed(pubkeys, messages, signatures)
    if n == 0:
        return False, b""
    st = ctypes.create_string_buffer(n)
    ok = _check(lib().mi355_bls_batch_verify_compressed(cache._h, pk, ms, sg, n, _rnd32(secureRandomBytes), st))
    return bool(ok), st.raw


def signSets(cache, secret_keys, messages):
    """Batch signer / input generator: per tuple publicFromSecret + coreSign
    (blst_min_pubkey_sig_core.nim:118-133, :230-251) on the device, VARIABLE TIME (test and bench inputs only).
    secret_keys: n x 32-byte little-endian scalars, messages: n x 32 bytes (lists or concatenated).
    -> (all_valid, n x 320-byte SignatureSet records, per-tuple status bytes: 1 = sk == 0 or sk >= r)."""
    sk = secret_keys if isinstance(secret_keys, (bytes, bytearray)) else b"".join(bytes(x) for x in secret_keys)
    ms = messages if isinstance(messages, (bytes, bytearray)) else b"".join(bytes(x) for x in messages)
    assert len(sk) % 32 == 0 and len(ms) == len(sk)
    n = len(sk) // 32
    if n == 0:
        return True, b"", b""
    out = ctypes.create_string_buffer(320 * n)
    st = ctypes.create_string_buffer(n)
    ok = _check(lib().mi355_bls_sign_sets(cache._h, bytes(sk), bytes(ms), n, out, st))
    return bool(ok), out.raw, st.raw


def signSets_device(cache, d_sks, d_msgs, n, d_out, stream=0):
    """Same with the scalars, messages and the output records resident in device memory (raw pointers)."""
    st = ctypes.create_string_buffer(max(n, 1))
    ok = _check(lib().mi355_bls_sign_sets_device(cache._h, d_sks, d_msgs, n, d_out, stream, st))
    return bool(ok), st.raw[:n]


class MultiSignatureSet:
    """bls_batch_verifier.nim:47-106: signatures that all pertain to the same 32-byte message."""

    def __init__(self, pubkeys, message, signatures):
        pubkeys, signatures = list(pubkeys), list(signatures)
        assert len(pubkeys) == len(signatures) and len(pubkeys) > 0          # doAssert :80-81
        self.pubkeys, self.message, self.signatures = pubkeys, bytes(message), signatures

    @classmethod
    def init(cls, pubkeys, message=None, signatures=None):
        if message is None:                       # init(sigset: SignatureSet)  (:89-94)
            pk, msg, sig = pubkeys
            return cls([pk], msg, [sig])
        return cls(pubkeys, message, signatures)

    def add(self, sigset):
        pk, msg, sig = sigset
        assert bytes(msg) == self.message                                     # doAssert :97
        self.pubkeys.append(pk)
        self.signatures.append(sig)

    def combine(self, cache, secureRandomBytes):
        """-> SignatureSet (pubkey96, message32, signature192)."""
        n = len(self.pubkeys)
        out_pk, out_sig = ctypes.create_string_buffer(96), ctypes.create_string_buffer(192)
        _check(lib().mi355_bls_combine(cache._h, _rnd32(secureRandomBytes), b"".join(self.pubkeys), b"".join(self.signatures), n, out_pk, out_sig))
        return (out_pk.raw, self.message, out_sig.raw)


def aggregateVerify(cache, publicKeys, messages, signature):
    """bls_sig_min_pubkey.nim:153-174: one aggregate signature over distinct (public key, message) pairs.
    Length mismatch or an empty list -> False."""
    pks, msgs = list(publicKeys), [bytes(x) for x in messages]
    if len(pks) != len(msgs) or len(pks) == 0:
        return False
    if any(len(p) != 96 for p in pks) or len(signature) not in (192, 288):
        raise ValueError("public keys are 96-byte; the signature a 192-byte Signature (affine) or a 288-byte AggregateSignature (blst_p2)")
    offs = [0]
    for x in msgs:
        offs.append(offs[-1] + len(x))
    arr = (ctypes.c_uint32 * len(offs))(*offs)
    fn = lib().mi355_bls_aggregate_verify if len(signature) == 192 else lib().mi355_bls_aggregate_verify_p2
    return bool(_check(fn(cache._h, b"".join(pks), b"".join(msgs) or b"\0", arr, len(pks), bytes(signature))))


class ContextCoreAggregateVerify:
    """blst_min_pubkey_sig_core.nim:305-414, the streaming form of aggregateVerify: init() / update(publicKey, message) -> bool /
    finish(signature) -> bool, on a BatchedBLSVerifierCache's device context (mi355_bls_aggv_*)."""

    def __init__(self, cache):
        self._c = cache

    def init(self):
        _check(lib().mi355_bls_aggv_init(self._c._h))

    def update(self, publicKey, message):
        if len(publicKey) != 96:
            raise ValueError("public keys are 96-byte blst_p1_affine images")
        m = bytes(message)
        return bool(_check(lib().mi355_bls_aggv_update(self._c._h, bytes(publicKey), m or None, len(m))))

    def finish(self, signature):
        """finish(signature: Signature or AggregateSignature) (core :357): 192-byte affine or 288-byte Jacobian image"""
        if len(signature) == 288:
            return bool(_check(lib().mi355_bls_aggv_finish_p2(self._c._h, bytes(signature))))
        if len(signature) != 192:
            raise ValueError("the signature is a 192-byte blst_p2_affine or a 288-byte blst_p2 image")
        return bool(_check(lib().mi355_bls_aggv_finish(self._c._h, bytes(signature))))


def aggregateVerifyStreaming(cache, publicKeys, messages, signature):
    """aggregateVerify written as the reference writes it (bls_sig_min_pubkey.nim:153-174): ctx.init, one update per pair, finish."""
    pks, msgs = list(publicKeys), list(messages)
    if len(pks) != len(msgs) or len(pks) == 0:
        return False
    ctx = ContextCoreAggregateVerify(cache)
    ctx.init()
    for pk, m in zip(pks, msgs):
        if not ctx.update(pk, m):
            return False
    return ctx.finish(signature)


def msm_shard_range(npoints, world, rank):
    first, count = ctypes.c_size_t(), ctypes.c_size_t()
    lib().mi355_bls_msm_shard_range(npoints, world, rank, ctypes.byref(first), ctypes.byref(count))
    return first.value, count.value


def p1s_mult_pippenger_multi(caches, points, scalars, nbits=255, g2=False):
    """blst_p1s_mult_pippenger (g2: blst_p2s) point-sharded over several devices from one host thread: caches[g] lives on device g."""
    ab = 192 if g2 else 96
    if len(points) % ab or len(scalars) % 32 or len(points) // ab != len(scalars) // 32:
        raise ValueError("points: n x %d bytes, scalars: n x 32 bytes" % ab)
    n = len(points) // ab
    arr = (ctypes.c_void_p * len(caches))(*[c._h for c in caches])
    pb = ctypes.create_string_buffer(bytes(points), len(points)) if n else None
    sb = ctypes.create_string_buffer(bytes(scalars), len(scalars)) if n else None
    pl = (ctypes.c_void_p * 2)(ctypes.addressof(pb) if n else None, None)
    sl = (ctypes.c_void_p * 2)(ctypes.addressof(sb) if n else None, None)
    out = ctypes.create_string_buffer(288 if g2 else 144)
    fn = lib().mi355_bls_p2s_mult_pippenger_multi if g2 else lib().mi355_bls_p1s_mult_pippenger_multi
    _check(fn(arr, len(caches), out, pl, n, sl, nbits))
    return out.raw


def p1s_mult_pippenger_multi_device(caches, d_points, n, d_scalars, nbits=255):
    """Same with shard g's arrays (msm_shard_range(n, len(caches), g)) already resident on device g: d_points[g], d_scalars[g]."""
    arr = (ctypes.c_void_p * len(caches))(*[c._h for c in caches])
    dp = (ctypes.c_void_p * len(caches))(*d_points)
    ds = (ctypes.c_void_p * len(caches))(*d_scalars)
    out = ctypes.create_string_buffer(144)
    _check(lib().mi355_bls_p1s_mult_pippenger_multi_device(arr, len(caches), out, dp, n, ds, nbits))
    return out.raw


def p1s_mult_pippenger_partial_device(cache, d_out, d_points, n, d_scalars, nbits=255, stream=0):
    """This device's partial of a point-sharded MSM, left at d_out (144 B, device memory) behind the rest of `stream`."""
    _check(lib().mi355_bls_p1s_mult_pippenger_partial_device(cache._h, d_out, d_points, n, d_scalars, nbits, stream))


def p1s_add(cache, parts, g2=False):
    """Sum of k blst_p1 (g2: blst_p2) Jacobian images (blst_p1_add_or_double, blst_abi.nim:278): the merge of MSM partials."""
    jb = 288 if g2 else 144
    buf = bytes(parts) if isinstance(parts, (bytes, bytearray, memoryview)) else b"".join(parts)
    if len(buf) % jb or not buf:
        raise ValueError("k x %d-byte Jacobian images" % jb)
    out = ctypes.create_string_buffer(jb)
    _check((lib().mi355_bls_p2s_add if g2 else lib().mi355_bls_p1s_add)(cache._h, out, buf, len(buf) // jb))
    return out.raw


def p1s_add_device(cache, d_parts, k, stride=144, stream=0):
    out = ctypes.create_string_buffer(144)
    _check(lib().mi355_bls_p1s_add_device(cache._h, out, d_parts, k, stride, stream))
    return out.raw
