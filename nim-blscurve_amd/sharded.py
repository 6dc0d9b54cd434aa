"""Multi-GPU batch verification: one process per GPU, shards by signature set, ONE exchange.

Mirrors the reference's parallel path across devices instead of threads
(blscurve/bls_batch_verifier.nim:296-371): the global batch is cut into B = min(n, numThreads)
chunks by parallel_chunks (parallel_chunks.nim:42-66); rank g owns a contiguous block of chunks
(its processSingleChunk work, :326-341), commits a local pairing state (with its own
(AggrSign_g, -G1) pair folded in), and the states are merged by an Fp12 product
(blst_pairing_merge, linear merge of :360-364) followed by one finalVerify (:371) on rank 0.

The exchange is an all_gather of 584 bytes per rank (576-byte Fp12 state + ok flag).  An Fp12
product is not an elementwise reduction, so it is not expressible as an RCCL all_reduce op; the
payload is latency-bound, xGMI bandwidth is irrelevant.
"""

STATE_BYTES = 576
BLOB_BYTES = 584


def shard_plan(n_total, num_threads, world):
    """[(chunk_lo, chunk_hi, first_tuple, tuple_count)] per rank; chunks are dealt in contiguous,
    balanced blocks (the same +-1 rule parallel_chunks uses for tuples)."""
    b = min(n_total, num_threads)
    base, rem = divmod(b, world)
    tb, tr = divmod(n_total, b) if b else (0, 0)

    def toff(c):
        return (tb + 1) * c if c < tr else tb * c + tr

    plan = []
    lo = 0
    for g in range(world):
        hi = lo + base + (1 if g < rem else 0)
        plan.append((lo, hi, toff(lo), toff(hi) - toff(lo)))
        lo = hi
    return plan


def batch_verify_sharded(cache, local_sets_ptr, n_total, rank, world, secureRandomBytes, all_gather, stream=0):
    """cache: BatchedBLSVerifierCache of this rank (numThreads = GLOBAL number of blinding chains);
    local_sets_ptr: device pointer to this rank's tuples (plan[rank] range); all_gather(bytes) ->
    list of every rank's bytes.  Returns the verdict on rank 0, None elsewhere.
    n_total == 0 -> False (bls_batch_verifier.nim:312-314)."""
    if n_total == 0:
        return False if rank == 0 else None
    lo, hi, first, count = shard_plan(n_total, cache.numThreads, world)[rank]
    if count > 0:
        state, ok = cache.shard_device(local_sets_ptr, n_total, lo, hi, secureRandomBytes, stream)
        blob = state + bytes([1 if ok else 0]) + bytes(7)
    else:                                   # more ranks than chunks: neutral element
        blob = bytes(STATE_BYTES) + bytes([2]) + bytes(7)
    blobs = all_gather(blob)
    if rank != 0:
        return None
    live = [b for b in blobs if b[STATE_BYTES] != 2]
    if not all(b[STATE_BYTES] == 1 for b in live):
        return False                        # some update() failed (infinity public key)
    return cache.finalverify_shards([b[:STATE_BYTES] for b in live])


def msm_shard_range(npoints, world, rank):
    """(first, count) of the points rank `rank` takes in a point-sharded MSM: balanced contiguous blocks
    (mi355_bls_msm_shard_range; SURVEY.md section 8(e), "MSM")."""
    base, rem = divmod(npoints, world)
    first = (base + 1) * rank if rank < rem else base * rank + rem
    return first, base + (1 if rank < rem else 0)


P1_BYTES = 144


def msm_sharded(local_partial, add_partials, npoints, rank, world, all_gather):
    """Point-sharded blst_p1s_mult_pippenger across ranks (benchmarks/bls12381_msm_g1.nim:47-59 shape over GPUs):
    local_partial(first, count) -> this rank's 144-byte blst_p1 partial (all zeros for an empty shard),
    all_gather(bytes) -> every rank's bytes, add_partials([bytes]) -> their sum (rank 0 only; blst_p1_add_or_double).
    Returns the blst_p1 result on rank 0, None elsewhere.  One exchange of 144 bytes per rank."""
    first, count = msm_shard_range(npoints, world, rank)
    part = local_partial(first, count) if count else bytes(P1_BYTES)
    parts = all_gather(part)
    if rank != 0:
        return None
    return add_partials(parts)
