/* C ABI of the MI355X-native BLS12-381 batch-verification path.
 *
 * Drop-in boundary: these entry points are what nim-blscurve's batch layer would bind (importc)
 * instead of driving BLST tuple by tuple.  Every function cites the reference interface it
 * replaces (paths relative to the nim-blscurve tree).  Plain pointers and sizes only.
 *
 * Data layouts are the reference's in-memory ones (BLST structs, Montgomery limbs R = 2^384,
 * little-endian u64 x 6 per Fp; blscurve/blst/blst_abi.nim:87-122):
 *   blst_p1_affine  96 B (x, y)          blst_p1  144 B (x, y, z)   Jacobian
 *   blst_p2_affine 192 B (x.c0,x.c1,y.c0,y.c1)   blst_p2 288 B      Jacobian
 *   blst_fp12      576 B
 *   SignatureSet   320 B = pubkey @0 (96) | message @96 (32) | signature @128 (192)
 *                  (blscurve/bls_batch_verifier.nim:34; affine infinity = all-zero bytes)
 *
 * Return convention: 1 = verified (Nim `true`), 0 = not verified (Nim `false`), negative = runtime
 * failure (no GPU, HIP error, capacity); the reference API only has bool, the Nim shim maps <0 to
 * a Defect.  The library never falls back to a CPU path.
 */
#ifndef BLSCURVE_MI355X_H
#define BLSCURVE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_BLS_SIGSET_BYTES 320
#define MI355_BLS_FP12_BYTES 576
#define MI355_BLS_P1_BYTES 144
#define MI355_BLS_P2_BYTES 288
#define MI355_BLS_BLOB_BYTES 640     /* device-resident shard blob: 576-byte state | u32 ok word | zero padding */

#define MI355_BLS_ERR_HIP (-1)       /* a HIP call failed; see mi355_bls_last_error() */
#define MI355_BLS_ERR_CAPACITY (-2)  /* n exceeds the context's capacity (never returned by the batchVerify / aggregateVerify entry points:
                                        they process larger inputs in slices) */
#define MI355_BLS_ERR_ARG (-3)

typedef struct mi355_bls_ctx mi355_bls_ctx;

/* BatchedBLSVerifierCache.init / init(tp) (bls_batch_verifier.nim:108-119): persistent device
 * workspace on HIP device `device`.  One context per concurrent caller, reusable across calls
 * (bls_batch_verifier.nim:389-391).  max_sets sizes the workspace (about 32 KB of HBM per set), it does NOT bound
 * input.len: like the reference's cache (per-thread contexts only, :108-119,:141) every batchVerify entry point accepts
 * any n - a batch (or shard) larger than max_sets is processed in ceil(n / max_sets) balanced slices on the same stream,
 * whose committed states are merged on the device (blst_pairing_merge semantics).  Size it for the batches you expect:
 * a slice that fills the chip (>= 65 536 sets) runs at full throughput. */
int mi355_bls_ctx_create(mi355_bls_ctx** out, int device, size_t max_sets);
void mi355_bls_ctx_destroy(mi355_bls_ctx* ctx);
const char* mi355_bls_last_error(void);
/* How the loaded library was built: "aligned=1 dpp_combine=off stamp=<sha256 of its sources>".  aligned=0 = built without the
 * instruction-alignment post-pass (BLS_NO_ALIGN=1; ~23 % lower issue rate of the multiply-add streams): a measurement taken with
 * such a library says so (bench.py prints the string and refuses aligned=0 unless asked).  Static string, never NULL. */
const char* mi355_bls_build_info(void);

/* HIP hardware queues.  HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
 * queue run strictly in turn.  Whole-chip batches do not care; a host that keeps MANY SMALL batches in flight (one context + stream
 * each; 4 096-tuple batches: 1.4 M verifications/s with 4 queues, 2.3 M/s with 8; more than 8 abort in the runtime) should run with
 * GPU_MAX_HW_QUEUES=8.  The variable is read when the HIP runtime initialises, so it is the HOST's to set - in its environment, or by
 * calling this helper ONCE from its main thread before anything in the process touches HIP (setenv is not thread-safe, and the
 * setting changes queue behaviour for every HIP user of the process: torch, RCCL).  The library never edits the environment on its
 * own.  Returns 1 if GPU_MAX_HW_QUEUES is now set (an existing value is kept), 0 if MI355_BLS_NO_ENV=1 forbids it or setenv failed. */
int mi355_bls_recommend_hw_queues(void);

/* Taskpool.numThreads analogue (bls_batch_verifier.nim:316): the number of blinding-scalar hash
 * chains ("virtual threads") B = min(n, num_threads) the parallel path splits a batch into; chunk c
 * is seeded SHA256(rnd || LE64(c)) exactly as processSingleChunk does (:333-336).  Default 4096. */
int mi355_bls_ctx_set_num_threads(mi355_bls_ctx* ctx, uint32_t num_threads);

/* Latency mode (on = 1, the default) or throughput mode (on = 0) of a context.
 * Latency mode shortens ONE call at the price of some extra lane-work: batches of up to ~11 000 sets (which do not fill the chip
 * with one lane per set) run their cofactor clearing and Miller lines on the lane-team engine, 16 lanes per set (4 096 sets:
 * 15 ms in round 1, 5.35 in round 5, 3.9 ms now), with [r]PK and the signature side on two fork streams beside the hashing;
 * whole-chip batches run the signature side and the Miller lines of its extra pairs on a second stream beside the hashing, so
 * that no nearly empty round of waves follows a full one; the partial line products are folded on the lane-cooperative Fp12
 * engine, whose workgroups have three waves in this mode.
 * Throughput mode does the least total work: for a caller that keeps several batches in flight (one context each), where
 * the nearly empty rounds of one batch overlap the wide kernels of another.  Verdicts and GT values are the same. */
int mi355_bls_ctx_set_cooperative(mi355_bls_ctx* ctx, int on);

/* batchVerifyParallel / batchVerify raw-pointer overloads (bls_batch_verifier.nim:296-302,420-426):
 * sets = n x 320-byte SignatureSet records in HOST memory, rnd = secureRandomBytes. n == 0 -> 0.
 * WHEN TO CALL IT: one blocking call costs about 2.3 ms for any n up to ~200 and 3.0 ms up to ~1 000 (latency-bound chains: hash-to-G2,
 * Miller walk, final exponentiation), 3.5 ms at 4 096, 12.4 ms at 65 536 (11 ms per batch when three are kept in flight).  A CPU BLST verifies a small
 * batch faster than that: below about 5 sets per host core (~80 sets on 16 cores; bench.py's `crossover`, INTEGRATION.md "When to call the
 * GPU") a host should keep its CPU path, as the Nim shim of INTEGRATION.md does (Mi355MinSets).  Many small batches at once:
 * mi355_bls_batch_verify_many. */
int mi355_bls_batch_verify(mi355_bls_ctx* ctx, const void* sets, size_t n, const uint8_t rnd[32]);

/* batchVerifySerial (bls_batch_verifier.nim:121-160): same check with the serial scalar chain
 * (seed = SHA256(rnd), one chain over the whole batch).  The chain is inherently sequential: it is computed on the calling
 * host thread (about 0.3 us per tuple) and uploaded; everything else runs on the device as in the parallel path. */
int mi355_bls_batch_verify_serial(mi355_bls_ctx* ctx, const void* sets, size_t n, const uint8_t rnd[32]);

/* Same as mi355_bls_batch_verify with the records already resident in device memory (HBM) and
 * work enqueued on `stream` (hipStream_t, may be NULL); synchronises the stream before returning. */
int mi355_bls_batch_verify_device(mi355_bls_ctx* ctx, const void* d_sets, size_t n, const uint8_t rnd[32], void* stream);

/* Asynchronous form of mi355_bls_batch_verify_device, for a caller that keeps several batches in flight from ONE host
 * thread (one context + one stream per batch in flight, cf. "one cache per concurrent caller",
 * bls_batch_verifier.nim:389-391): submit enqueues the whole verification on `stream` and returns at once (0, or a
 * negative error; n == 0 is an error here); wait blocks until that batch is done and returns its verdict (1 / 0) exactly
 * as mi355_bls_batch_verify_device would.  One batch per context at a time; d_sets must stay valid until wait returns;
 * rnd is consumed at submit.  `after` (optional): another context whose batch was submitted earlier; this batch then
 * starts when that batch has finished hashing and multiplying its public keys, so the batches in flight sit at different
 * stages and the serial tail of one (a few waves: step products, Horner, final exponentiation) always runs beside
 * whole-chip kernels of another - deterministic software pipelining; three contexts are enough. */
int mi355_bls_batch_submit_device(mi355_bls_ctx* ctx, const void* d_sets, size_t n, const uint8_t rnd[32], void* stream,
                                  mi355_bls_ctx* after);
int mi355_bls_batch_wait(mi355_bls_ctx* ctx);

/* Many independent batches in ONE device pass (no reference counterpart: the reference verifies one batch per call and gets its
 * concurrency from caller threads; on the device, small batches cannot fill the chip and the number of HIP hardware queues caps
 * how many calls run side by side).  sets: the tuples of batch 0, then batch 1, ... (counts[b] of them each, 320-byte records),
 * rnds: k x 32 bytes, one secureRandomBytes per batch; verdicts[b] receives what mi355_bls_batch_verify_once(batch b, rnd b,
 * num_threads) would return (an empty batch: 0).  Every tuple keeps the blinding scalar it has in its own batch (own chain
 * partition B = min(n_b, num_threads), serial chain for n_b < 3 or num_threads = 1, bls_batch_verifier.nim:440); the union is
 * verified at once, and the product of the k batch checks is one iff every batch verifies (up to the 2^-64 of the random linear
 * combination, the reference's own bound).  If it is not, or if the union exceeds the context's capacity, the batches are
 * verified one by one.  Returns 1 when every batch verified, 0 otherwise, negative on runtime failure.
 * REQUIREMENT for the merged pass: the k secureRandomBytes must be pairwise independent.  The library checks what it can: if any two
 * non-empty batches carry the SAME 32 bytes (their blinding chains would coincide and errors could cancel ACROSS batches, which k
 * separate calls would not allow), no merged pass is made and the batches are verified one by one - the verdicts stay those of k
 * separate calls, only the speed-up is lost.  Draw one fresh rnd per batch.
 * COST: a passing call is one whole-chip pass.  A call in which some batch fails costs that pass PLUS k single-batch calls (about
 * twice the latency; the calls are synchronous): an adversary who can place one bad signature per call forces the slow path for all
 * k batches - hosts that expect failures should keep k small or verify suspicious batches separately.
*/
int mi355_bls_batch_verify_many(mi355_bls_ctx* ctx, const void* sets, const size_t counts[], const uint8_t* rnds, size_t k, uint8_t verdicts[]);
int mi355_bls_batch_verify_many_device(mi355_bls_ctx* ctx, const void* d_sets, const size_t counts[], const uint8_t* rnds, size_t k, uint8_t verdicts[],
                                       void* stream);

/* Multi-GPU sharding (replaces processSingleChunk + merge, bls_batch_verifier.nim:326-369).
 * The global batch of n_total sets is cut into B = min(n_total, num_threads) chunks by
 * parallel_chunks (parallel_chunks.nim:42-66); this call processes chunks [chunk_lo, chunk_hi),
 * whose records start at d_sets (device memory), and returns the shard's committed pairing state:
 *   out_fp12 = prod_{i in shard} ML(H(m_i), [r_i]PK_i) * ML(sum [r_i]S_i, -G1)   (576 B, pre final-exp)
 *   *out_ok  = 0 if an update failed (infinity public key), else 1.
 * The signature-side pair is folded per shard, so merging shards is an Fp12 product only
 * (blst_pairing_merge, blst_abi.nim:508). */
int mi355_bls_batch_shard_device(mi355_bls_ctx* ctx, const void* d_sets, size_t n_total, uint32_t chunk_lo, uint32_t chunk_hi,
                                 const uint8_t rnd[32], void* stream, uint8_t out_fp12[576], int* out_ok);
/* asynchronous form (see mi355_bls_batch_submit_device / mi355_bls_batch_wait) */
int mi355_bls_batch_shard_submit_device(mi355_bls_ctx* ctx, const void* d_sets, size_t n_total, uint32_t chunk_lo, uint32_t chunk_hi,
                                        const uint8_t rnd[32], void* stream, mi355_bls_ctx* after);
int mi355_bls_batch_shard_wait(mi355_bls_ctx* ctx, uint8_t out_fp12[576], int* out_ok);

/* merge + finalVerify (blst_min_pubkey_sig_core.nim:657-672): product of k shard states (host
 * memory, k x 576 B), one final exponentiation on the device, == 1. */
int mi355_bls_finalverify_shards(mi355_bls_ctx* ctx, const uint8_t* fp12s, size_t k);

/* Device-resident exchange for one-process-per-GPU callers (the multi-GPU bench): every shard submit also writes the shard's
 * state + ok word into the context's blob buffer (MI355_BLS_BLOB_BYTES, device memory) on the submit's stream.  The caller
 * gathers the blobs of all ranks with a collective on device buffers (RCCL all_gather) and hands the gathered buffer to
 * finalverify_blobs: merge (blst_pairing_merge, core :657-666) + finalVerify (:670-672) on k blobs `stride_bytes` apart in
 * DEVICE memory, enqueued on `stream`; finalverify_wait blocks and returns the verdict (1 only if every shard's ok word is 1
 * and the product is one).  Nothing but the verdict word crosses PCIe.  (The collective may be enqueued right behind the
 * submit on the same stream; the bench waits for the shard first - mi355_bls_batch_shard_wait, a host synchronisation - because
 * torch issues collectives on a stream of its own, where one that waits for a whole batch blocks a hardware queue.)
 * finalverify_blobs keeps its verdict word and GT value apart from the batch path's, so a context may take its next shard
 * while a merge submitted on it is still in flight on another stream. */
int mi355_bls_ctx_shard_blob_device(mi355_bls_ctx* ctx, void** d_blob);
/* Redirect the blob to the caller's own device buffer (MI355_BLS_BLOB_BYTES, 16-byte aligned; e.g. the send buffer of the
 * collective, so no copy is needed); NULL restores the context's internal buffer. */
int mi355_bls_ctx_set_shard_blob_device(mi355_bls_ctx* ctx, void* d_blob);
int mi355_bls_finalverify_blobs_submit_device(mi355_bls_ctx* ctx, const void* d_blobs, size_t k, size_t stride_bytes, void* stream);
int mi355_bls_finalverify_wait(mi355_bls_ctx* ctx);

/* Which chunks device `rank` of `world` takes: contiguous balanced blocks of the B = min(n_total, num_threads) chunks,
 * and the tuple range they cover. */
int mi355_bls_shard_plan(size_t n_total, uint32_t num_threads, uint32_t world, uint32_t rank, uint32_t* chunk_lo, uint32_t* chunk_hi,
                         size_t* first, size_t* count);

/* batchVerifyParallel across several GPUs of one node from ONE host thread (bls_batch_verifier.nim:296-371 with devices in
 * place of taskpool threads): ctxs[g] is a context on device g (all with the same num_threads; a shard larger than its
 * context's capacity is sliced, the largest shard is mi355_bls_shard_plan(...).count of rank 0); shard g = the chunk block
 * mi355_bls_shard_plan gives rank g.  The plan and every context are validated before anything is enqueued; the caller's
 * host range is page-locked for the call so that all shards' copies and kernels are enqueued asynchronously (:342-357) and
 * device g does not wait for device g - 1's staging; the 576-byte states return through pinned host memory, ctxs[0] merges
 * them and runs the one final exponentiation (:360-371).  If enqueuing a shard fails, the shards already submitted are waited
 * for before the error is returned, so every context stays usable.  `sets`: n x 320 B in host memory; the _device form takes
 * d_sets[g] = shard g's records already resident on device g.  n == 0 -> 0. */
int mi355_bls_batch_verify_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* sets, size_t n, const uint8_t rnd[32]);
int mi355_bls_batch_verify_multi_device(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* const d_sets[], size_t n, const uint8_t rnd[32]);

/* The cache-less overloads batchVerifyParallel(tp, input, rnd) / batchVerify(tp, input, rnd) (bls_batch_verifier.nim:399-416,
 * :475-495), which build a BatchedBLSVerifierCache per call: here a process-wide default context (HIP device
 * $MI355_BLS_DEVICE, default 0) created on first use and regrown on demand; calls are serialised by a mutex.
 * num_threads = tp.numThreads; dispatch as batchVerify: parallel iff num_threads > 1 and n >= 3, else the serial chain. */
int mi355_bls_batch_verify_once(const void* sets, size_t n, const uint8_t rnd[32], uint32_t num_threads);
void mi355_bls_default_ctx_release(void);

/* Helper: tuple range [*first, *first + *count) covered by chunks [chunk_lo, chunk_hi) of a batch of
 * n_total sets split into num_threads chunks (parallel_chunks.nim:42-66). */
void mi355_bls_chunk_range(size_t n_total, uint32_t num_threads, uint32_t chunk_lo, uint32_t chunk_hi, size_t* first, size_t* count);

/* aggregateAll on G1 (blst_min_pubkey_sig_core.nim:179-195: blst_p1_from_affine + a serial loop of
 * blst_p1_add_or_double_affine): sum of n blst_p1_affine points -> blst_p1 (Jacobian, 144 B; the
 * caller finishes with blst_p1_to_affine as the reference's `finish` does).  Host or device input. */
int mi355_bls_g1_aggregate(mi355_bls_ctx* ctx, const void* pks, size_t n, uint8_t out_p1[144]);
int mi355_bls_g1_aggregate_device(mi355_bls_ctx* ctx, const void* d_pks, size_t n, void* stream, uint8_t out_p1[144]);
/* aggregateAll on signatures (genAggregatorProcedures(AggregateSignature, Signature, p2), blst_min_pubkey_sig_core.nim:179-195,211):
 * sigs: n x 192 B blst_p2_affine (infinity = all zero contributes nothing), out_p2: the sum as blst_p2 (Jacobian, 288 B) - what
 * mi355_bls_aggregate_verify_p2 / mi355_bls_aggv_finish_p2 take.  n == 0: MI355_BLS_ERR_ARG (the reference's openArray form
 * returns false before touching its output). */
int mi355_bls_g2_aggregate(mi355_bls_ctx* ctx, const void* sigs, size_t n, uint8_t out_p2[288]);
int mi355_bls_g2_aggregate_device(mi355_bls_ctx* ctx, const void* d_sigs, size_t n, void* stream, uint8_t out_p2[288]);

/* fastAggregateVerify(publicKeys, message, signature) (bls_sig_min_pubkey.nim:234-258): aggregate the
 * n public keys on the device, then coreVerifyNoGroupCheck (core :269-297): e(agg, H(msg)) == e(G1, sig).
 * pks: n x 96 B blst_p1_affine, sig: 192 B blst_p2_affine (host memory), msg_len <= 4096.
 * n == 0 -> 0; aggregate at infinity -> 0. */
int mi355_bls_fast_aggregate_verify(mi355_bls_ctx* ctx, const void* pks, size_t n, const uint8_t* msg, size_t msg_len, const void* sig);
int mi355_bls_fast_aggregate_verify_device(mi355_bls_ctx* ctx, const void* d_pks, size_t n, const uint8_t* msg, size_t msg_len,
                                           const void* sig, void* stream);
/* The same with the keys sharded over the GPUs of one node: ctxs[g] on device g sums the contiguous block of
 * keys mi355_bls_msm_shard_range(n, ngpu, g) gives it (aggregateAll, core :179-195), the 144-byte partial sums are added on ctxs[0]
 * (blst_p1_add_or_double), which runs the one pairing check.  pks: n x 96 B in host memory. */
int mi355_bls_fast_aggregate_verify_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* pks, size_t n, const uint8_t* msg,
                                          size_t msg_len, const void* sig);
/* coreVerifyNoGroupCheck on an AggregatePublicKey the caller already holds (core :269-297): agg_p1 = blst_p1 (Jacobian, 144 B), e.g.
 * mi355_bls_p1s_add of the per-rank mi355_bls_g1_aggregate_device partial sums of a key-sharded fastAggregateVerify (one process per
 * GPU).  Aggregate at infinity -> 0. */
int mi355_bls_verify_aggregate(mi355_bls_ctx* ctx, const uint8_t agg_p1[144], const uint8_t* msg, size_t msg_len, const void* sig);

/* blst_p1s_mult_pippenger (blst_abi.nim:336-340; blst+nim.h:70-72; call sites benchmarks/bls12381_msm_g1.nim:50-59,
 * blst_min_pubkey_sig_core.nim:629): ret = sum_i [k_i mod 2^nbits] P_i as blst_p1 (Jacobian, 144 B).
 * points[0] -> npoints contiguous blst_p1_affine, scalars[0] -> npoints 32-byte little-endian scalars, both
 * NULL-terminated pointer lists in HOST memory exactly as the reference passes them; nbits in 1..256.
 * The device workspace is (re)sized inside the context on first use, so the scratch size is 0. */
size_t mi355_bls_p1s_mult_pippenger_scratch_sizeof(size_t npoints);
int mi355_bls_p1s_mult_pippenger(mi355_bls_ctx* ctx, uint8_t ret_p1[144], const void* const points[], size_t npoints,
                                 const uint8_t* const scalars[], size_t nbits);
/* EXACTLY blst_p1s_mult_pippenger / blst_p1s_mult_pippenger_scratch_sizeof (blst+nim.h:70-72; blst_abi.nim:336-340): same
 * argument list and void result, so `importc: "mi355_p1s_mult_pippenger"` replaces the BLST symbol with no other change at
 * the call sites (benchmarks/bls12381_msm_g1.nim:47-59; blst_min_pubkey_sig_core.nim:629-636).  blst's conventions:
 *   points[] / scalars[]: list[0] points at element 0; each following element is taken from the next list entry if that is
 *     non-NULL, else it follows the previous element in memory ([ptr, NULL] = one contiguous array; npoints pointers = one
 *     per element);
 *   scalars are little-endian, (nbits + 7) / 8 bytes each (32 for nbits = 255, 8 for the u64 scalars of `combine`);
 *   ret: blst_p1 (Jacobian, 144 B); scratch: ignored (the workspace lives on the device; scratch_sizeof returns 8).
 * Runs on the process-wide default context.  A runtime failure (no GPU, HIP error) cannot be returned through a void
 * signature: the function prints the error and aborts. */
size_t mi355_p1s_mult_pippenger_scratch_sizeof(size_t npoints);
void mi355_p1s_mult_pippenger(void* ret, const void* const points[], size_t npoints, const uint8_t* const scalars[], size_t nbits, void* scratch);
/* EXACTLY blst_p2s_mult_pippenger / ..._scratch_sizeof (blst+nim.h:90-92; blst_abi.nim:358-362; call site
 * blst_min_pubkey_sig_core.nim:639-646): the same on G2 (points: blst_p2_affine, 192 B; ret: blst_p2, 288 B). */
size_t mi355_p2s_mult_pippenger_scratch_sizeof(size_t npoints);
void mi355_p2s_mult_pippenger(void* ret, const void* const points[], size_t npoints, const uint8_t* const scalars[], size_t nbits, void* scratch);
/* context forms on G2 (32-byte scalar images, host or device arrays), as the mi355_bls_p1s_* pair */
int mi355_bls_p2s_mult_pippenger(mi355_bls_ctx* ctx, uint8_t ret_p2[288], const void* const points[], size_t npoints,
                                 const uint8_t* const scalars[], size_t nbits);
int mi355_bls_p2s_mult_pippenger_device(mi355_bls_ctx* ctx, uint8_t ret_p2[288], const void* d_points, size_t npoints,
                                        const void* d_scalars, size_t nbits, void* stream);

/* same with both arrays resident in device memory */
int mi355_bls_p1s_mult_pippenger_device(mi355_bls_ctx* ctx, uint8_t ret_p1[144], const void* d_points, size_t npoints,
                                        const void* d_scalars, size_t nbits, void* stream);

/* Point-sharded MSM across the GPUs of one node (blst_p1s_mult_pippenger, blst_abi.nim:336-340, over devices; the bench shape
 * benchmarks/bls12381_msm_g1.nim:47-59): device g computes the full-width partial sum of points [first_g, first_g + count_g)
 * (mi355_bls_msm_shard_range: balanced contiguous blocks), the 144-byte partials are added (blst_p1_add_or_double,
 * blst_abi.nim:278).
 *   _multi         one host thread, ctxs[g] on device g, host arrays as mi355_bls_p1s_mult_pippenger (32-byte scalar images);
 *                  all shards are enqueued before any is waited for, ctxs[0] adds the partials
 *   _multi_device  d_points[g] / d_scalars[g] = shard g's arrays already resident on device g
 *   _partial_device + p1s_add(_device)   one process per GPU: every rank leaves its partial in device memory (the send buffer
 *                  of an RCCL all_gather), rank 0 adds the gathered partials (k x stride_bytes in device memory, or k x 144 B in
 *                  host memory) */
void mi355_bls_msm_shard_range(size_t npoints, uint32_t world, uint32_t rank, size_t* first, size_t* count);
int mi355_bls_p1s_mult_pippenger_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p1[144], const void* const points[], size_t npoints,
                                       const uint8_t* const scalars[], size_t nbits);
int mi355_bls_p2s_mult_pippenger_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p2[288], const void* const points[], size_t npoints,
                                       const uint8_t* const scalars[], size_t nbits);
int mi355_bls_p1s_mult_pippenger_multi_device(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p1[144], const void* const d_points[], size_t npoints,
                                              const void* const d_scalars[], size_t nbits);
int mi355_bls_p1s_mult_pippenger_partial_device(mi355_bls_ctx* ctx, void* d_out_p1, const void* d_points, size_t npoints, const void* d_scalars,
                                                size_t nbits, void* stream);
int mi355_bls_p1s_add(mi355_bls_ctx* ctx, uint8_t ret_p1[144], const uint8_t* parts, size_t k);
int mi355_bls_p2s_add(mi355_bls_ctx* ctx, uint8_t ret_p2[288], const uint8_t* parts, size_t k);
int mi355_bls_p1s_add_device(mi355_bls_ctx* ctx, uint8_t ret_p1[144], const void* d_parts, size_t k, size_t stride_bytes, void* stream);

/* Batched PublicKey.fromBytes / Signature.fromBytes (blscurve/blst/bls_sig_io.nim:42-58, 81-99) on the device:
 * n compressed public keys (48 B each), 32-byte messages and compressed signatures (96 B each), ZCash format.
 * Per tuple: blst_p1_uncompress, "public key is not infinity", blst_p1_affine_in_g1, blst_p2_uncompress,
 * blst_p2_affine_in_g2 (an infinity signature is allowed).  status[i] (optional, n bytes): 0 ok, 1 bad pk
 * encoding / x >= p / not on the curve, 2 pk not in G1, 3 pk infinity, 4 bad sig encoding, 5 sig not in G2.
 * out_sets (optional): n x 320-byte SignatureSet records (failed tuples are zeroed).
 * Returns 1 when every tuple deserialised, 0 when some did not, negative on runtime failure. */
int mi355_bls_deserialize_sets(mi355_bls_ctx* ctx, const uint8_t* pks48, const uint8_t* msgs32, const uint8_t* sigs96, size_t n,
                               void* out_sets, uint8_t* status);
int mi355_bls_deserialize_sets_device(mi355_bls_ctx* ctx, const void* d_pks48, const void* d_msgs32, const void* d_sigs96, size_t n,
                                      void* stream, void* out_sets, uint8_t* status);

/* The other forms of fromBytes (bls_sig_io.nim:42-121), selected per side by `flags`:
 *   MI355_BLS_DESER_PK_UNCOMPRESSED   keys are 96-byte images   -> blst_p1_deserialize (:88-91)
 *   MI355_BLS_DESER_SIG_UNCOMPRESSED  signatures are 192-byte images -> blst_p2_deserialize (:49-52)
 *   MI355_BLS_DESER_KNOWN_ON_CURVE    fromBytesKnownOnCurve (:60-79, :101-121): no subgroup checks (the infinity public key is
 *                                     still rejected)
 * blst_pN_deserialize [blst-upstream]: top bits of byte 0 = 000 uncompressed big-endian coordinates (G2: x.c1, x.c0, y.c1, y.c0),
 * each < p, point on the curve; 1xx a compressed encoding in the first half; 01x infinity (0x40 followed by zeros only).
 * pks: n x 48 or 96 bytes, sigs: n x 96 or 192 bytes; status codes and results as mi355_bls_deserialize_sets. */
#define MI355_BLS_DESER_PK_UNCOMPRESSED 1u
#define MI355_BLS_DESER_SIG_UNCOMPRESSED 2u
#define MI355_BLS_DESER_KNOWN_ON_CURVE 4u
int mi355_bls_deserialize_sets_ex(mi355_bls_ctx* ctx, const uint8_t* pks, const uint8_t* msgs32, const uint8_t* sigs, size_t n, uint32_t flags,
                                  void* out_sets, uint8_t* status);
int mi355_bls_deserialize_sets_ex_device(mi355_bls_ctx* ctx, const void* d_pks, const void* d_msgs32, const void* d_sigs, size_t n, uint32_t flags,
                                         void* stream, void* out_sets, uint8_t* status);

/* fromBytes for every tuple followed by batchVerify (bls_batch_verifier.nim:420-495), all on the device:
 * the wire format (176 B per tuple) is the only thing that crosses PCIe.  Returns 0 if any tuple fails to
 * deserialise (the caller would not have obtained a SignatureSet) or the batch does not verify. */
int mi355_bls_batch_verify_compressed(mi355_bls_ctx* ctx, const uint8_t* pks48, const uint8_t* msgs32, const uint8_t* sigs96, size_t n,
                                      const uint8_t rnd[32], uint8_t* status);
int mi355_bls_batch_verify_compressed_device(mi355_bls_ctx* ctx, const void* d_pks48, const void* d_msgs32, const void* d_sigs96, size_t n,
                                             const uint8_t rnd[32], void* stream, uint8_t* status);
float mi355_bls_last_deser_ms(mi355_bls_ctx* ctx);   /* duration of the deserialisation kernel of the last compressed call */

/* combine(secureRandomBytes, publicKeys, signatures) (blst_min_pubkey_sig_core.nim:570-647; called by
 * MultiSignatureSet.combine, bls_batch_verifier.nim:100-106): linear combination of n signatures on ONE message.
 * Scalars: chain seeded with rnd itself, taking the u64 words 3,2,1,0 of each SHA-256 output, zeros skipped
 * (:588-606); out_pk = sum [s_i]PK_i and out_sig = sum [s_i]S_i as affine BLST images: two 64-bit Pippenger runs on the
 * device (G1 and G2 bucket kernels, the reference's blst_p1s/p2s_mult_pippenger calls, :629-646) + finish.  The scalar chain
 * is sequential SHA-256: it runs on the calling host thread.  n == 1: passthrough; n == 0: MI355_BLS_ERR_ARG (the reference asserts).
 * pks: n x 96 B, sigs: n x 192 B, host memory.  Returns 0 on success. */
int mi355_bls_combine(mi355_bls_ctx* ctx, const uint8_t rnd[32], const void* pks, const void* sigs, size_t n, uint8_t out_pk[96],
                      uint8_t out_sig[192]);

/* aggregateVerify(publicKeys, messages, signature) (bls_sig_min_pubkey.nim:153-199; ContextCoreAggregateVerify,
 * blst_min_pubkey_sig_core.nim:305-414): e(G1, sig) == prod_i e(pk_i, H(m_i)) for n (public key, message) pairs
 * with messages of arbitrary length: message i = msgs[msg_offsets[i] .. msg_offsets[i+1]) (n + 1 offsets).
 * pks: n x 96 B, sig: 192 B, host memory.  n == 0 -> 0; infinity public key -> 0.  The proofs of possession
 * must have been checked by the caller, as for the reference's two-argument overloads.  Any n: inputs beyond the
 * context's capacity are processed in slices. */
int mi355_bls_aggregate_verify(mi355_bls_ctx* ctx, const void* pks, const uint8_t* msgs, const uint32_t* msg_offsets, size_t n,
                               const void* sig);
/* the same with the signature as an AggregateSignature (blst_p2, Jacobian, 288 B): finish(signature: AggregateSignature),
 * blst_min_pubkey_sig_core.nim:357 - converted to affine on the device (blst_p2_to_affine) before the pairing */
int mi355_bls_aggregate_verify_p2(mi355_bls_ctx* ctx, const void* pks, const uint8_t* msgs, const uint32_t* msg_offsets, size_t n,
                                  const void* sig_p2);

/* The streaming form, ContextCoreAggregateVerify.init / update / finish (blst_min_pubkey_sig_core.nim:321-414; driven by
 * bls_sig_min_pubkey.nim:127-199 for the AoS and SoA overloads): init resets the context's pair list; update(publicKey,
 * message) appends one pair and returns 1, or 0 for the infinity public key (the reference's update returns false:
 * BLST_PK_IS_INFINITY) after which finish returns 0; finish(signature) = commit + finalVerify: one device call over the
 * collected pairs, returns the verdict (0 when no pair was added) and consumes the context (init again before reuse).
 * pk: 96-byte blst_p1_affine, sig: 192-byte blst_p2_affine, any message length. */
int mi355_bls_aggv_init(mi355_bls_ctx* ctx);
int mi355_bls_aggv_update(mi355_bls_ctx* ctx, const void* pk, const uint8_t* msg, size_t msg_len);
int mi355_bls_aggv_finish(mi355_bls_ctx* ctx, const void* sig);
/* finish(signature: AggregateSignature) (core :357): sig_p2 = blst_p2, Jacobian, 288 B.  update returns MI355_BLS_ERR_CAPACITY (and
 * appends nothing) once the collected messages would exceed 4 GiB (offsets are 32-bit). */
int mi355_bls_aggv_finish_p2(mi355_bls_ctx* ctx, const void* sig_p2);

/* Batch signer / input generator.  Per tuple i, from a 32-byte little-endian secret scalar
 * (blst_scalar image, SecretKey, blst_min_pubkey_sig_core.nim:43-66) and a 32-byte message:
 *   publicFromSecret (core :118-133): sk == 0 or sk >= r -> status[i] = 1 and a zeroed record; pk = affine([sk]G1)
 *   coreSign (core :230-251) with the signature DST (bls_sig_min_pubkey.nim:31): sig = affine([sk]H(msg))
 * written as the 320-byte SignatureSet record (pk, msg, sig).  Returns 1 when every key was valid, 0 otherwise.
 * The scalar multiplications are VARIABLE TIME: this exists to synthesise test and bench inputs (the reference
 * does the same with sign() in benchmarks/bls_signature.nim:258-268), never to sign with real keys. */
int mi355_bls_sign_sets(mi355_bls_ctx* ctx, const uint8_t* sks32, const uint8_t* msgs32, size_t n, void* out_sets, uint8_t* status);
int mi355_bls_sign_sets_device(mi355_bls_ctx* ctx, const void* d_sks32, const void* d_msgs32, size_t n, void* d_out_sets, void* stream,
                               uint8_t* status);

/* Stage outputs of the LAST batch call on this context, for parity tests (no reference
 * counterpart: BLST keeps these inside blst_pairing).  `what`:
 *   0: blinding scalars r_i           n x 8 B  (LE u64)
 *   1: H(m_i) Jacobian                n x 288 B (blst_p2)
 *   2: [r_i]PK_i Jacobian             n x 144 B (blst_p1)
 *   3: sum [r_i]S_i Jacobian          288 B
 *   4: GT after final exponentiation  576 B   (f^(3 (p^12-1)/r))
 *   5: Miller-loop product before final exponentiation 576 B */
int mi355_bls_fetch_stage(mi355_bls_ctx* ctx, int what, void* out, size_t out_bytes);

/* Kernel timing of the last batch call, ms per stage measured with HIP events on the call's stream:
 * out[0..7] = blinding, hash_to_g2, pk_mul, sig_mul+sum, miller_lines, line_products, final, total. */
int mi355_bls_last_timings(mi355_bls_ctx* ctx, float out[8]);
/* Per-kernel split of the two-kernel stages of the last batch call (ms): k_hash_map, k_hash_clear (hash_to_g2),
 * k_lineprod, k_lineprod2 (line_products). */
int mi355_bls_last_kernel_timings(mi355_bls_ctx* ctx, float out[4]);

/* TEST HOOK: out[i] = clear_cofactor(q0_i + q1_i) (RFC 9380 G.3, the last stage of hash-to-G2) for n <= max_sets pairs of blst_p2 images
 * (2 x 288 B per pair, host memory), computed by the kernels the batch path would launch for this context (k_hash_clear on a throughput-mode context, the lane-team engine on a
 * latency-mode one) - lets the tests put points through them that no hash produces
 * (the point at infinity, equal or opposite points: the cases its incomplete addition formulas flag and recompute). */
int mi355_bls_debug_g2_clear_cofactor(mi355_bls_ctx* ctx, const uint8_t* in_pairs, size_t n, uint8_t* out_p2);

/* TEST HOOK: out_p2 = hash_to_G2(msg, dst) (blst_p2 image, Jacobian) computed by the one-message kernel of fastAggregateVerify / verify under ANY
 * domain separation tag (1 .. 64 bytes): holds the device against published hash-to-curve vectors (RFC 9380 J.10.1), whose DST is not the scheme's. */
int mi355_bls_debug_hash_to_g2(mi355_bls_ctx* ctx, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t out_p2[288]);

/* Test hooks (no reference counterpart).  debug_fail_next_enqueue: the next batch / shard enqueue on this context fails with
 * MI355_BLS_ERR_HIP before touching the device (exercises the multi-device driver's clean-up path).  debug_multi_enqueue_us:
 * host time in microseconds, counted from the start of the last mi355_bls_batch_verify_multi* call of this thread, at which each
 * device's shard was handed to its stream (the start skew between devices); returns the number of devices recorded. */
int mi355_bls_debug_fail_next_enqueue(mi355_bls_ctx* ctx);
size_t mi355_bls_debug_multi_enqueue_us(float* out, size_t cap);
/* Batches submitted and not yet waited for, over all contexts of the process: what the library looks at when it chooses between the
 * low-latency and the least-work fold of the line products (a batch enqueued while this is zero has the chip to itself).  The tests
 * check that submit / wait / destroy keep it balanced. */
int mi355_bls_debug_batches_in_flight(void);
/* Which fold of the per-lane line products the LAST batch / shard call on this context enqueued: 1 = the low-latency form on the Fp12 engine
 * (k_fold: latency-mode contexts, and any call enqueued while no other batch of the process was in flight), 0 = the least-work form
 * (k_lineprod2: throughput-mode contexts with other batches in flight).  Same GT bytes either way; profiles and the bench line record it so
 * that a kernel mix can be attributed.  Negative: bad argument. */
int mi355_bls_last_fold_form(mi355_bls_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
