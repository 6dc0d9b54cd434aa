/* A compiled (non-Python) caller of the C ABI: plain C, linked against libblscurve_mi355x.so, using only
 * include/blscurve_mi355x.h - the way a Nim `importc` binding reaches the library.
 * Reads tests/golden/cabi_fixture.bin (layout: tests/golden/gen_cabi_bin.py) and prints one line per check:
 *   batchVerify(cache) / batchVerifySerial / cache-less batchVerify on the golden n17 batch      -> verdicts (expect 1)
 *   the same on the golden forged_among_many batch                                                -> verdicts (expect 0)
 *   blst_p1s_mult_pippenger-shaped MSM (n = 32): [ptr, NULL] lists and one-pointer-per-element lists -> blst_p1 hex
 *   the same batches through a context sized for 5 sets (sliced inside the library) and through two contexts (multi-device driver)
 *   the MSM point-sharded over three contexts, and as partials + mi355_bls_p1s_add
 *   ContextCoreAggregateVerify-style streaming: init / update x n / finish with a wrong signature -> 0 (the batches carry one
 *   signature per tuple, not an aggregate), update with the infinity key -> 0
 * The pytest wrapper (tests/test_gpu_cabi.py) compares the verdicts and canonicalises the points against the fixtures. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blscurve_mi355x.h"

static unsigned char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* b = (unsigned char*)malloc(*len);
    if (fread(b, 1, *len, f) != *len) { perror("fread"); exit(2); }
    fclose(f);
    return b;
}
static unsigned rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
static void hex(const char* tag, const unsigned char* p, size_t n) {
    printf("%s ", tag);
    for (size_t i = 0; i < n; i++) printf("%02x", p[i]);
    printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s cabi_fixture.bin\n", argv[0]); return 2; }
    size_t len;
    unsigned char* f = slurp(argv[1], &len);
    if (len < 12 || memcmp(f, "MI355CAB", 8)) { fprintf(stderr, "bad fixture\n"); return 2; }
    const unsigned char* p = f + 8;
    mi355_bls_ctx* ctx = NULL;
    int rc = mi355_bls_ctx_create(&ctx, 0, 64);               /* BatchedBLSVerifierCache.init */
    if (rc) { fprintf(stderr, "ctx_create: %d %s\n", rc, mi355_bls_last_error()); return 1; }
    mi355_bls_ctx_set_num_threads(ctx, 4);                    /* Taskpool.new(numThreads = 4), tests/t_batch_verifier.nim:63 */
    for (int k = 0; k < 2; k++) {
        unsigned n = rd32(p);
        const unsigned char* sets = p + 4;
        const unsigned char* rnd = sets + 320 * (size_t)n;
        p = rnd + 32;
        printf("batch%d n %u\n", k, n);
        printf("batch%d parallel %d\n", k, mi355_bls_batch_verify(ctx, sets, n, rnd));
        printf("batch%d serial %d\n", k, mi355_bls_batch_verify_serial(ctx, sets, n, rnd));
        printf("batch%d once %d\n", k, mi355_bls_batch_verify_once(sets, n, rnd, 4));
        /* a cache smaller than the batch: the reference accepts any input.len (bls_batch_verifier.nim:141) */
        mi355_bls_ctx* tiny = NULL;
        if (mi355_bls_ctx_create(&tiny, 0, 5)) { fprintf(stderr, "ctx_create(5): %s\n", mi355_bls_last_error()); return 1; }
        mi355_bls_ctx_set_num_threads(tiny, 4);
        printf("batch%d sliced %d\n", k, mi355_bls_batch_verify(tiny, sets, n, rnd));
        printf("batch%d sliced_serial %d\n", k, mi355_bls_batch_verify_serial(tiny, sets, n, rnd));
        mi355_bls_ctx* pair[2] = {ctx, tiny};
        printf("batch%d multi %d\n", k, mi355_bls_batch_verify_multi(pair, 2, sets, n, rnd));
        /* streaming aggregateVerify over the batch's (pk, msg) pairs with tuple 0's signature: not the aggregate -> 0 */
        mi355_bls_aggv_init(tiny);
        int upd = 1;
        for (unsigned i = 0; i < n; i++) upd &= mi355_bls_aggv_update(tiny, sets + 320 * (size_t)i, sets + 320 * (size_t)i + 96, 32);
        printf("batch%d aggv_updates %d\n", k, upd);
        printf("batch%d aggv_finish %d\n", k, mi355_bls_aggv_finish(tiny, sets + 128));
        {   /* aggregateAll on the batch's signatures (core :179-195,211) -> AggregateSignature (blst_p2, 288 B) -> finish(AggregateSignature) (core :357) */
            unsigned char* sg = (unsigned char*)malloc(192 * (size_t)n);
            unsigned char agg[288];
            for (unsigned i = 0; i < n; i++) memcpy(sg + 192 * (size_t)i, sets + 320 * (size_t)i + 128, 192);
            printf("batch%d g2_aggregate %d\n", k, mi355_bls_g2_aggregate(tiny, sg, n, agg));
            mi355_bls_aggv_init(tiny);
            for (unsigned i = 0; i < n; i++) mi355_bls_aggv_update(tiny, sets + 320 * (size_t)i, sets + 320 * (size_t)i + 96, 32);
            printf("batch%d aggv_p2_finish %d\n", k, mi355_bls_aggv_finish_p2(tiny, agg));
            mi355_bls_g2_aggregate(tiny, sg, n - 1, agg);                       /* one signature short: not the aggregate */
            mi355_bls_aggv_init(tiny);
            for (unsigned i = 0; i < n; i++) mi355_bls_aggv_update(tiny, sets + 320 * (size_t)i, sets + 320 * (size_t)i + 96, 32);
            printf("batch%d aggv_p2_short %d\n", k, mi355_bls_aggv_finish_p2(tiny, agg));
            free(sg);
        }
        unsigned char zero_pk[96] = {0};
        mi355_bls_aggv_init(tiny);
        printf("batch%d aggv_inf_update %d\n", k, mi355_bls_aggv_update(tiny, zero_pk, sets + 96, 32));
        printf("batch%d aggv_inf_finish %d\n", k, mi355_bls_aggv_finish(tiny, sets + 128));
        mi355_bls_ctx_destroy(tiny);
    }
    printf("empty %d\n", mi355_bls_batch_verify(ctx, f, 0, f + 8));
    {   /* both batches (and an empty one between them) in ONE device pass: per-batch verdicts */
        const unsigned char* q = f + 8;
        unsigned n0 = rd32(q), n1 = rd32(q + 4 + 320 * (size_t)n0 + 32);
        const unsigned char* s0 = q + 4;
        const unsigned char* s1 = s0 + 320 * (size_t)n0 + 32 + 4;
        unsigned char* all = (unsigned char*)malloc(320 * (size_t)(n0 + n1));
        memcpy(all, s0, 320 * (size_t)n0);
        memcpy(all + 320 * (size_t)n0, s1, 320 * (size_t)n1);
        unsigned char rnds[96], verdicts[3] = {9, 9, 9};
        memcpy(rnds, s0 + 320 * (size_t)n0, 32);
        memset(rnds + 32, 7, 32);
        memcpy(rnds + 64, s1 + 320 * (size_t)n1, 32);
        size_t counts[3] = {n0, 0, n1};
        int rcm = mi355_bls_batch_verify_many(ctx, all, counts, rnds, 3, verdicts);
        printf("many %d%d%d%d\n", rcm, verdicts[0], verdicts[1], verdicts[2]);
        free(all);
    }
    unsigned np = rd32(p), nbits = rd32(p + 4);
    const unsigned char* pts = p + 8;
    const unsigned char* sc = pts + 96 * (size_t)np;
    unsigned char ret[144];
    void* scratch = malloc(mi355_p1s_mult_pippenger_scratch_sizeof(np));      /* benchmarks/bls12381_msm_g1.nim:50 */
    const void* pl[2] = {pts, NULL};                                           /* "Weird API with double indirection" (:53) */
    const unsigned char* sl[2] = {sc, NULL};
    mi355_p1s_mult_pippenger(ret, pl, np, sl, nbits, scratch);
    hex("msm_contiguous", ret, 144);
    const void** pe = (const void**)malloc(np * sizeof(void*));
    const unsigned char** se = (const unsigned char**)malloc(np * sizeof(void*));
    for (unsigned i = 0; i < np; i++) { pe[i] = pts + 96 * (size_t)i; se[i] = sc + 32 * (size_t)i; }
    mi355_p1s_mult_pippenger(ret, pe, np, se, nbits, scratch);
    hex("msm_pointer_list", ret, 144);
    if (mi355_bls_p1s_mult_pippenger(ctx, ret, pl, np, sl, nbits)) { fprintf(stderr, "msm(ctx): %s\n", mi355_bls_last_error()); return 1; }
    hex("msm_ctx", ret, 144);
    /* point-sharded over three contexts (devices in a real deployment), and the one-process-per-GPU merge of partials */
    mi355_bls_ctx* three[3] = {ctx, NULL, NULL};
    if (mi355_bls_ctx_create(&three[1], 0, 8) || mi355_bls_ctx_create(&three[2], 0, 8)) { fprintf(stderr, "ctx_create: %s\n", mi355_bls_last_error()); return 1; }
    if (mi355_bls_p1s_mult_pippenger_multi(three, 3, ret, pl, np, sl, nbits)) { fprintf(stderr, "msm multi: %s\n", mi355_bls_last_error()); return 1; }
    hex("msm_multi", ret, 144);
    unsigned char parts[3 * 144];
    for (unsigned g = 0; g < 3; g++) {
        size_t first, count;
        mi355_bls_msm_shard_range(np, 3, g, &first, &count);
        const void* gpl[2] = {pts + 96 * first, NULL};
        const unsigned char* gsl[2] = {sc + 32 * first, NULL};
        if (mi355_bls_p1s_mult_pippenger(three[g], parts + 144 * g, gpl, count, gsl, nbits)) { fprintf(stderr, "msm part: %s\n", mi355_bls_last_error()); return 1; }
    }
    if (mi355_bls_p1s_add(ctx, ret, parts, 3)) { fprintf(stderr, "p1s_add: %s\n", mi355_bls_last_error()); return 1; }
    hex("msm_partials_added", ret, 144);
    mi355_bls_ctx_destroy(three[1]);
    mi355_bls_ctx_destroy(three[2]);
    free(scratch); free(pe); free(se);
    mi355_bls_ctx_destroy(ctx);
    mi355_bls_default_ctx_release();
    free(f);
    return 0;
}
