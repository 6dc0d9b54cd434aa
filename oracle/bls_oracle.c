/* CPU restatement (plain C, 6 x u64 limbs, unsigned __int128) of the reference's batch-verification
 * algorithm.  TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Never linked into or called by the product.
 *
 * It follows the REFERENCE's structure, not the product's:
 *   - per-thread pairing context {GT, AggrSign, queue of N_MAX=8 (Q,P) pairs}       blst_abi.nim:147-178
 *   - init / update / commit / merge / finalverify                                  blst_min_pubkey_sig_core.nim:476-568,649-672
 *   - parallel_chunks split over OpenMP threads, linear merge                       bls_batch_verifier.nim:296-371, parallel_chunks.nim:42-66
 *   - update = blinding chain SHA256, AggrSign += [r]S, H = hash_to_G2 -> AFFINE,
 *     P = [r]PK -> AFFINE, queue; every 8 pairs a Miller loop with SHARED squarings [blst-upstream]
 * BLST itself (vendor/blst, un-checked-out submodule; v0.3.13 linked at core :609) is absent; its
 * arithmetic is restated from RFC 9380 / the optimal-ate pairing, with constants generated from
 * the KAT-pinned Python oracle (oracle_constants.h).  Deliberately different from the product:
 * 64-bit limbs, affine hash/pk outputs (field inversions), RFC-style SSWU with explicit inversion
 * and generic Fp2 square root, rational isogeny map, classic per-pair Miller accumulation.
 * Pinned by tests/test_c_oracle.py against the golden fixtures (which carry the reference's KATs).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 a0, a1, a2; } fp6;
typedef struct { fp6 c0, c1; } fp12;
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp x, y, z; } g1j;
typedef struct { fp2 x, y; int inf; } g2a;
typedef struct { fp2 x, y, z; } g2j;

/* ------------------------------------------------------------------ Fp */
static fp fp_c(const uint64_t* k) { fp r; memcpy(r.l, k, 48); return r; }
static fp fp_zero(void) { fp r; memset(&r, 0, sizeof r); return r; }
static fp fp_one(void) { return fp_c(K_ONE); }
static int fp_is_zero(const fp* a) { uint64_t t = 0; for (int i = 0; i < 6; i++) t |= a->l[i]; return t == 0; }
static int fp_eq(const fp* a, const fp* b) { return memcmp(a, b, 48) == 0; }

static void fp_cond_sub(fp* r, uint64_t carry) {
    uint64_t d[6]; u128 b = 0;
    for (int i = 0; i < 6; i++) { u128 v = (u128)r->l[i] - K_P[i] - (uint64_t)b; d[i] = (uint64_t)v; b = (v >> 64) & 1; }
    if (carry || !b) memcpy(r->l, d, 48);
}
static fp fp_add(const fp* a, const fp* b) {
    fp r; u128 c = 0;
    for (int i = 0; i < 6; i++) { c += (u128)a->l[i] + b->l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    fp_cond_sub(&r, (uint64_t)c); return r;
}
static fp fp_sub(const fp* a, const fp* b) {
    fp r; u128 br = 0;
    for (int i = 0; i < 6; i++) { u128 v = (u128)a->l[i] - b->l[i] - (uint64_t)br; r.l[i] = (uint64_t)v; br = (v >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + K_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static fp fp_neg(const fp* a) { fp z = fp_zero(); return fp_sub(&z, a); }
static fp fp_mul(const fp* a, const fp* b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * K_N0;
        c = ((u128)m * K_P[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) { c += (u128)m * K_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
    }
    fp r; memcpy(r.l, t, 48); fp_cond_sub(&r, t[6]); return r;
}
static fp fp_sqr(const fp* a) { return fp_mul(a, a); }
static fp fp_pow(const fp* a, const uint64_t* e, int nlimbs) {
    fp r = fp_one(); int started = 0;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        if (started) r = fp_sqr(&r);
        if ((e[i >> 6] >> (i & 63)) & 1) { r = started ? fp_mul(&r, a) : *a; started = 1; }
    }
    return r;
}
static fp fp_inv(const fp* a) { return fp_pow(a, K_PM2, 6); }
static int fp_sqrt(fp* r, const fp* a) { fp s = fp_pow(a, K_PP1D4, 6); fp q = fp_sqr(&s); *r = s; return fp_eq(&q, a); }
static fp fp_from_mont(const fp* a) { fp one = fp_zero(); one.l[0] = 1; return fp_mul(a, &one); }
static fp fp_to_mont(const fp* a) { fp rr = fp_c(K_RR); return fp_mul(a, &rr); }
static fp fp_small(uint64_t v) { fp a = fp_zero(); a.l[0] = v; return fp_to_mont(&a); }

/* ------------------------------------------------------------------ Fp2 */
static fp2 f2_c(const uint64_t* k) { fp2 r; r.c0 = fp_c(k); r.c1 = fp_c(k + 6); return r; }
static fp2 f2_zero(void) { fp2 r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
static fp2 f2_one(void) { fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
static int f2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int f2_eq(const fp2* a, const fp2* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static fp2 f2_add(const fp2* a, const fp2* b) { fp2 r; r.c0 = fp_add(&a->c0, &b->c0); r.c1 = fp_add(&a->c1, &b->c1); return r; }
static fp2 f2_sub(const fp2* a, const fp2* b) { fp2 r; r.c0 = fp_sub(&a->c0, &b->c0); r.c1 = fp_sub(&a->c1, &b->c1); return r; }
static fp2 f2_neg(const fp2* a) { fp2 r; r.c0 = fp_neg(&a->c0); r.c1 = fp_neg(&a->c1); return r; }
static fp2 f2_conj(const fp2* a) { fp2 r; r.c0 = a->c0; r.c1 = fp_neg(&a->c1); return r; }
static fp2 f2_dbl(const fp2* a) { return f2_add(a, a); }
static fp2 f2_mul(const fp2* a, const fp2* b) {
    fp t0 = fp_mul(&a->c0, &b->c0), t1 = fp_mul(&a->c1, &b->c1);
    fp sa = fp_add(&a->c0, &a->c1), sb = fp_add(&b->c0, &b->c1), s = fp_mul(&sa, &sb);
    fp2 r; r.c0 = fp_sub(&t0, &t1); s = fp_sub(&s, &t0); r.c1 = fp_sub(&s, &t1); return r;
}
static fp2 f2_sqr(const fp2* a) {
    fp s = fp_add(&a->c0, &a->c1), d = fp_sub(&a->c0, &a->c1), t = fp_mul(&a->c0, &a->c1);
    fp2 r; r.c0 = fp_mul(&s, &d); r.c1 = fp_add(&t, &t); return r;
}
static fp2 f2_mul_fp(const fp2* a, const fp* b) { fp2 r; r.c0 = fp_mul(&a->c0, b); r.c1 = fp_mul(&a->c1, b); return r; }
static fp2 f2_mul_xi(const fp2* a) { fp2 r; r.c0 = fp_sub(&a->c0, &a->c1); r.c1 = fp_add(&a->c0, &a->c1); return r; }
static fp2 f2_inv(const fp2* a) {
    fp n0 = fp_sqr(&a->c0), n1 = fp_sqr(&a->c1), n = fp_add(&n0, &n1), ni = fp_inv(&n);
    fp2 r; r.c0 = fp_mul(&a->c0, &ni); fp t = fp_mul(&a->c1, &ni); r.c1 = fp_neg(&t); return r;
}
static int f2_is_square(const fp2* a) {
    fp n0 = fp_sqr(&a->c0), n1 = fp_sqr(&a->c1), n = fp_add(&n0, &n1);
    if (fp_is_zero(&n)) return 1;
    fp l = fp_pow(&n, K_PM1D2, 6), one = fp_one();
    return fp_eq(&l, &one);
}
/* generic square root in Fp2 (complex method); returns 0 if none */
static int f2_sqrt(fp2* r, const fp2* a) {
    if (f2_is_zero(a)) { *r = f2_zero(); return 1; }
    fp two = fp_small(2), half = fp_inv(&two);
    if (fp_is_zero(&a->c1)) {
        fp s;
        if (fp_sqrt(&s, &a->c0)) { r->c0 = s; r->c1 = fp_zero(); return 1; }
        fp na = fp_neg(&a->c0); fp_sqrt(&s, &na); r->c0 = fp_zero(); r->c1 = s; return 1;
    }
    fp n0 = fp_sqr(&a->c0), n1 = fp_sqr(&a->c1), nn = fp_add(&n0, &n1), n;
    if (!fp_sqrt(&n, &nn)) return 0;
    fp d = fp_add(&a->c0, &n); d = fp_mul(&d, &half);
    fp x0;
    if (!fp_sqrt(&x0, &d)) { d = fp_sub(&a->c0, &n); d = fp_mul(&d, &half); if (!fp_sqrt(&x0, &d)) return 0; }
    fp tx = fp_add(&x0, &x0), txi = fp_inv(&tx);
    r->c0 = x0; r->c1 = fp_mul(&a->c1, &txi);
    fp2 chk = f2_sqr(r);
    return f2_eq(&chk, a);
}
static int f2_sgn0(const fp2* a) {
    fp x0 = fp_from_mont(&a->c0), x1 = fp_from_mont(&a->c1);
    int s0 = x0.l[0] & 1, z0 = fp_is_zero(&x0), s1 = x1.l[0] & 1;
    return s0 | (z0 & s1);
}

/* ------------------------------------------------------------------ Fp6 / Fp12 */
static fp6 f6_add(const fp6* a, const fp6* b) { fp6 r; r.a0 = f2_add(&a->a0, &b->a0); r.a1 = f2_add(&a->a1, &b->a1); r.a2 = f2_add(&a->a2, &b->a2); return r; }
static fp6 f6_sub(const fp6* a, const fp6* b) { fp6 r; r.a0 = f2_sub(&a->a0, &b->a0); r.a1 = f2_sub(&a->a1, &b->a1); r.a2 = f2_sub(&a->a2, &b->a2); return r; }
static fp6 f6_neg(const fp6* a) { fp6 r; r.a0 = f2_neg(&a->a0); r.a1 = f2_neg(&a->a1); r.a2 = f2_neg(&a->a2); return r; }
static fp6 f6_mul_v(const fp6* a) { fp6 r; r.a0 = f2_mul_xi(&a->a2); r.a1 = a->a0; r.a2 = a->a1; return r; }
static fp6 f6_mul(const fp6* a, const fp6* b) {   /* schoolbook: 9 products */
    fp2 p00 = f2_mul(&a->a0, &b->a0), p01 = f2_mul(&a->a0, &b->a1), p02 = f2_mul(&a->a0, &b->a2);
    fp2 p10 = f2_mul(&a->a1, &b->a0), p11 = f2_mul(&a->a1, &b->a1), p12 = f2_mul(&a->a1, &b->a2);
    fp2 p20 = f2_mul(&a->a2, &b->a0), p21 = f2_mul(&a->a2, &b->a1), p22 = f2_mul(&a->a2, &b->a2);
    fp6 r; fp2 t;
    t = f2_add(&p12, &p21); t = f2_mul_xi(&t); r.a0 = f2_add(&p00, &t);
    t = f2_mul_xi(&p22); r.a1 = f2_add(&p01, &p10); r.a1 = f2_add(&r.a1, &t);
    r.a2 = f2_add(&p02, &p11); r.a2 = f2_add(&r.a2, &p20);
    return r;
}
static fp6 f6_inv(const fp6* a) {
    fp2 t, c0, c1, c2;
    c0 = f2_sqr(&a->a0); t = f2_mul(&a->a1, &a->a2); t = f2_mul_xi(&t); c0 = f2_sub(&c0, &t);
    c1 = f2_sqr(&a->a2); c1 = f2_mul_xi(&c1); t = f2_mul(&a->a0, &a->a1); c1 = f2_sub(&c1, &t);
    c2 = f2_sqr(&a->a1); t = f2_mul(&a->a0, &a->a2); c2 = f2_sub(&c2, &t);
    fp2 d = f2_mul(&a->a2, &c1), e = f2_mul(&a->a1, &c2); d = f2_add(&d, &e); d = f2_mul_xi(&d);
    e = f2_mul(&a->a0, &c0); d = f2_add(&d, &e);
    fp2 di = f2_inv(&d);
    fp6 r; r.a0 = f2_mul(&c0, &di); r.a1 = f2_mul(&c1, &di); r.a2 = f2_mul(&c2, &di); return r;
}
static fp12 f12_one(void) { fp12 r; memset(&r, 0, sizeof r); r.c0.a0 = f2_one(); return r; }
static fp12 f12_mul(const fp12* a, const fp12* b) {
    fp6 t0 = f6_mul(&a->c0, &b->c0), t1 = f6_mul(&a->c1, &b->c1), t2 = f6_mul(&a->c0, &b->c1), t3 = f6_mul(&a->c1, &b->c0);
    fp12 r; fp6 v = f6_mul_v(&t1); r.c0 = f6_add(&t0, &v); r.c1 = f6_add(&t2, &t3); return r;
}
static fp12 f12_sqr(const fp12* a) { return f12_mul(a, a); }
static fp12 f12_conj(const fp12* a) { fp12 r; r.c0 = a->c0; r.c1 = f6_neg(&a->c1); return r; }
static fp12 f12_inv(const fp12* a) {
    fp6 s0 = f6_mul(&a->c0, &a->c0), s1 = f6_mul(&a->c1, &a->c1), v = f6_mul_v(&s1), d = f6_sub(&s0, &v), di = f6_inv(&d);
    fp12 r; r.c0 = f6_mul(&a->c0, &di); fp6 t = f6_mul(&a->c1, &di); r.c1 = f6_neg(&t); return r;
}
static int f12_is_one(const fp12* a) { fp12 o = f12_one(); return memcmp(a, &o, sizeof o) == 0; }
/* a^p : flat w^i coefficient i gets conj() * gamma_i ; tower slots c0.(a0,a1,a2)=w^0,2,4 c1.(..)=w^1,3,5 */
static fp12 f12_frob(const fp12* a) {
    fp12 r; fp2 t, g;
    r.c0.a0 = f2_conj(&a->c0.a0);
    t = f2_conj(&a->c1.a0); g = f2_c(K_FROB1); r.c1.a0 = f2_mul(&t, &g);
    t = f2_conj(&a->c0.a1); g = f2_c(K_FROB2); r.c0.a1 = f2_mul(&t, &g);
    t = f2_conj(&a->c1.a1); g = f2_c(K_FROB3); r.c1.a1 = f2_mul(&t, &g);
    t = f2_conj(&a->c0.a2); g = f2_c(K_FROB4); r.c0.a2 = f2_mul(&t, &g);
    t = f2_conj(&a->c1.a2); g = f2_c(K_FROB5); r.c1.a2 = f2_mul(&t, &g);
    return r;
}
/* f * (l0 + l1 v + l2 v w) */
static fp12 f12_mul_line(const fp12* f, const fp2* l0, const fp2* l1, const fp2* l2) {
    fp12 l; memset(&l, 0, sizeof l); l.c0.a0 = *l0; l.c0.a1 = *l1; l.c1.a1 = *l2;
    return f12_mul(f, &l);
}

/* ------------------------------------------------------------------ curves (Jacobian, a = 0) */
#define X_ABS 0xd201000000010000ull
#define DEF_CURVE(G, F, FZERO, FONE, FISZ, FADD, FSUB, FMUL, FSQR, FNEG, FEQ, FINV)                                          \
    static G##j G##_inf(void) { G##j r; r.x = FZERO(); r.y = FZERO(); r.z = FZERO(); return r; }                                 \
    static int G##_is_inf(const G##j* p) { return FISZ(&p->z); }                                                                \
    static G##j G##_from_aff(const G##a* p) { G##j r; if (p->inf) return G##_inf(); r.x = p->x; r.y = p->y; r.z = FONE(); return r; } \
    static G##j G##_dbl(const G##j* p) {                                                                                        \
        if (G##_is_inf(p)) return *p;                                                                                           \
        F A = FSQR(&p->x), B = FSQR(&p->y), C = FSQR(&B), t = FADD(&p->x, &B), D = FSQR(&t);                                     \
        D = FSUB(&D, &A); D = FSUB(&D, &C); D = FADD(&D, &D);                                                                   \
        F E = FADD(&A, &A); E = FADD(&E, &A); F Fq = FSQR(&E);                                                                  \
        G##j r; F D2 = FADD(&D, &D); r.x = FSUB(&Fq, &D2);                                                                       \
        F C8 = FADD(&C, &C); C8 = FADD(&C8, &C8); C8 = FADD(&C8, &C8);                                                           \
        t = FSUB(&D, &r.x); t = FMUL(&E, &t); r.y = FSUB(&t, &C8);                                                               \
        t = FMUL(&p->y, &p->z); r.z = FADD(&t, &t); return r;                                                                    \
    }                                                                                                                           \
    static G##j G##_add(const G##j* p, const G##j* q) {                                                                          \
        if (G##_is_inf(p)) return *q;                                                                                           \
        if (G##_is_inf(q)) return *p;                                                                                           \
        F z1z1 = FSQR(&p->z), z2z2 = FSQR(&q->z), u1 = FMUL(&p->x, &z2z2), u2 = FMUL(&q->x, &z1z1);                               \
        F s1 = FMUL(&p->y, &q->z); s1 = FMUL(&s1, &z2z2); F s2 = FMUL(&q->y, &p->z); s2 = FMUL(&s2, &z1z1);                       \
        F h = FSUB(&u2, &u1), rr = FSUB(&s2, &s1);                                                                               \
        if (FISZ(&h)) { if (FISZ(&rr)) return G##_dbl(p); return G##_inf(); }                                                    \
        F hh = FSQR(&h), hhh = FMUL(&h, &hh), v = FMUL(&u1, &hh);                                                                \
        G##j r; F t = FSQR(&rr); t = FSUB(&t, &hhh); F v2 = FADD(&v, &v); r.x = FSUB(&t, &v2);                                    \
        t = FSUB(&v, &r.x); t = FMUL(&rr, &t); F u = FMUL(&s1, &hhh); r.y = FSUB(&t, &u);                                         \
        t = FMUL(&p->z, &q->z); r.z = FMUL(&t, &h); return r;                                                                    \
    }                                                                                                                           \
    static G##j G##_neg(const G##j* p) { G##j r = *p; r.y = FNEG(&p->y); return r; }                                             \
    static G##j G##_mul(const G##j* p, const uint8_t* k_le, int nbits) {                                                         \
        G##j acc = G##_inf();                                                                                                   \
        for (int i = nbits - 1; i >= 0; i--) { acc = G##_dbl(&acc); if ((k_le[i >> 3] >> (i & 7)) & 1) acc = G##_add(&acc, p); } \
        return acc;                                                                                                             \
    }                                                                                                                           \
    static G##a G##_to_aff(const G##j* p) {                                                                                      \
        G##a r; memset(&r, 0, sizeof r);                                                                                        \
        if (G##_is_inf(p)) { r.inf = 1; return r; }                                                                             \
        F zi = FINV(&p->z), zi2 = FSQR(&zi), zi3 = FMUL(&zi2, &zi); r.x = FMUL(&p->x, &zi2); r.y = FMUL(&p->y, &zi3); return r;   \
    }
DEF_CURVE(g1, fp, fp_zero, fp_one, fp_is_zero, fp_add, fp_sub, fp_mul, fp_sqr, fp_neg, fp_eq, fp_inv)
DEF_CURVE(g2, fp2, f2_zero, f2_one, f2_is_zero, f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_eq, f2_inv)

static void u64_le(uint8_t* o, uint64_t v) { for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (8 * i)); }
static g2j g2_mul_u64(const g2j* p, uint64_t k) { uint8_t b[8]; u64_le(b, k); return g2_mul(p, b, 64); }
static g1j g1_mul_u64(const g1j* p, uint64_t k) { uint8_t b[8]; u64_le(b, k); return g1_mul(p, b, 64); }

/* blst memory images */
static fp ld_fp(const uint8_t* p) { fp r; memcpy(r.l, p, 48); return r; }
static void st_fp(uint8_t* p, const fp* a) { memcpy(p, a->l, 48); }
static g1a ld_g1a(const uint8_t* p) { g1a r; r.x = ld_fp(p); r.y = ld_fp(p + 48); r.inf = fp_is_zero(&r.x) && fp_is_zero(&r.y); return r; }
static g2a ld_g2a(const uint8_t* p) {
    g2a r; r.x.c0 = ld_fp(p); r.x.c1 = ld_fp(p + 48); r.y.c0 = ld_fp(p + 96); r.y.c1 = ld_fp(p + 144);
    r.inf = f2_is_zero(&r.x) && f2_is_zero(&r.y); return r;
}
static void st_g1a(uint8_t* p, const g1a* a) { if (a->inf) { memset(p, 0, 96); return; } st_fp(p, &a->x); st_fp(p + 48, &a->y); }
static void st_g2a(uint8_t* p, const g2a* a) {
    if (a->inf) { memset(p, 0, 192); return; }
    st_fp(p, &a->x.c0); st_fp(p + 48, &a->x.c1); st_fp(p + 96, &a->y.c0); st_fp(p + 144, &a->y.c1);
}
static void st_f12(uint8_t* p, const fp12* a) {
    const fp2* c[6] = {&a->c0.a0, &a->c0.a1, &a->c0.a2, &a->c1.a0, &a->c1.a1, &a->c1.a2};
    for (int i = 0; i < 6; i++) { st_fp(p + 96 * i, &c[i]->c0); st_fp(p + 96 * i + 48, &c[i]->c1); }
}

/* ------------------------------------------------------------------ SHA-256 */
static const uint32_t SK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t n; } sha_t;
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(uint32_t* h, const uint8_t* p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4]; f = h[5]; g = h[6]; hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + SK[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha_init(sha_t* s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, 32); s->n = 0;
}
static void sha_update(sha_t* s, const uint8_t* p, size_t len) {
    for (size_t i = 0; i < len; i++) { s->buf[s->n & 63] = p[i]; s->n++; if ((s->n & 63) == 0) sha_block(s->h, s->buf); }
}
static void sha_final(sha_t* s, uint8_t out[32]) {
    uint64_t bits = s->n * 8; uint8_t pad = 0x80; sha_update(s, &pad, 1); pad = 0;
    while ((s->n & 63) != 56) sha_update(s, &pad, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(s, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}
void oracle_sha256(const uint8_t* m, size_t n, uint8_t out[32]) { sha_t s; sha_init(&s); sha_update(&s, m, n); sha_final(&s, out); }

/* ------------------------------------------------------------------ hash to G2 (RFC 9380) */
static void expand_xmd(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, size_t n) {
    uint8_t b0[32], bi[32], z[64] = {0}, t[3], dl = (uint8_t)dlen; sha_t s;
    size_t ell = (n + 31) / 32;
    sha_init(&s); sha_update(&s, z, 64); sha_update(&s, msg, mlen);
    t[0] = (uint8_t)(n >> 8); t[1] = (uint8_t)n; t[2] = 0; sha_update(&s, t, 3); sha_update(&s, dst, dlen); sha_update(&s, &dl, 1); sha_final(&s, b0);
    for (size_t i = 1; i <= ell; i++) {
        uint8_t x[32], ib = (uint8_t)i;
        for (int j = 0; j < 32; j++) x[j] = i == 1 ? b0[j] : (uint8_t)(b0[j] ^ bi[j]);
        sha_init(&s); sha_update(&s, x, 32); sha_update(&s, &ib, 1); sha_update(&s, dst, dlen); sha_update(&s, &dl, 1); sha_final(&s, bi);
        size_t off = 32 * (i - 1), c = n - off < 32 ? n - off : 32; memcpy(out + off, bi, c);
    }
}
/* 64 big-endian bytes mod p -> Montgomery */
static fp fp_from_be64(const uint8_t* b) {
    fp acc = fp_zero(), k256 = fp_small(256);
    for (int i = 0; i < 64; i++) { acc = fp_mul(&acc, &k256); fp d = fp_small(b[i]); acc = fp_add(&acc, &d); }
    return acc;
}
static g2a sswu(const fp2* u) {
    fp2 A = f2_c(K_SSWU_A), B = f2_c(K_SSWU_B), Z = f2_c(K_SSWU_Z), one = f2_one();
    fp2 u2 = f2_sqr(u), zu2 = f2_mul(&Z, &u2), tv1 = f2_sqr(&zu2); tv1 = f2_add(&tv1, &zu2);
    fp2 x1, t, ai = f2_inv(&A), nb = f2_neg(&B);
    if (f2_is_zero(&tv1)) { t = f2_mul(&Z, &A); t = f2_inv(&t); x1 = f2_mul(&B, &t); }
    else { t = f2_inv(&tv1); t = f2_add(&one, &t); x1 = f2_mul(&nb, &ai); x1 = f2_mul(&x1, &t); }
    fp2 gx = f2_sqr(&x1); gx = f2_mul(&gx, &x1); t = f2_mul(&A, &x1); gx = f2_add(&gx, &t); gx = f2_add(&gx, &B);
    g2a r; r.inf = 0;
    if (f2_is_square(&gx)) { r.x = x1; f2_sqrt(&r.y, &gx); }
    else {
        fp2 x2 = f2_mul(&zu2, &x1), g2 = f2_sqr(&x2); g2 = f2_mul(&g2, &x2); t = f2_mul(&A, &x2); g2 = f2_add(&g2, &t); g2 = f2_add(&g2, &B);
        r.x = x2; f2_sqrt(&r.y, &g2);
    }
    if (f2_sgn0(u) != f2_sgn0(&r.y)) r.y = f2_neg(&r.y);
    return r;
}
static fp2 horner(const fp2* x, const uint64_t* const* k, int deg) {
    fp2 acc = f2_c(k[deg]);
    for (int i = deg - 1; i >= 0; i--) { acc = f2_mul(&acc, x); fp2 c = f2_c(k[i]); acc = f2_add(&acc, &c); }
    return acc;
}
static g2a iso3(const g2a* p) {   /* rational map, RFC 9380 appendix E.3 */
    static const uint64_t* const XN[4] = {K_ISO_XN0, K_ISO_XN1, K_ISO_XN2, K_ISO_XN3};
    static const uint64_t* const XD[3] = {K_ISO_XD0, K_ISO_XD1, K_ISO_XD2};
    static const uint64_t* const YN[4] = {K_ISO_YN0, K_ISO_YN1, K_ISO_YN2, K_ISO_YN3};
    static const uint64_t* const YD[4] = {K_ISO_YD0, K_ISO_YD1, K_ISO_YD2, K_ISO_YD3};
    g2a r; memset(&r, 0, sizeof r);
    if (p->inf) { r.inf = 1; return r; }
    fp2 xn = horner(&p->x, XN, 3), xd = horner(&p->x, XD, 2), yn = horner(&p->x, YN, 3), yd = horner(&p->x, YD, 3);
    if (f2_is_zero(&xd) || f2_is_zero(&yd)) { r.inf = 1; return r; }
    fp2 xdi = f2_inv(&xd), ydi = f2_inv(&yd);
    r.x = f2_mul(&xn, &xdi); r.y = f2_mul(&yn, &ydi); r.y = f2_mul(&r.y, &p->y); return r;
}
static g2j g2_psi(const g2j* p) {
    fp2 cx = f2_c(K_PSI_CX), cy = f2_c(K_PSI_CY); g2j r; fp2 t;
    t = f2_conj(&p->x); r.x = f2_mul(&t, &cx); t = f2_conj(&p->y); r.y = f2_mul(&t, &cy); r.z = f2_conj(&p->z); return r;
}
static g2j g2_mul_x(const g2j* p) { g2j t = g2_mul_u64(p, X_ABS); return g2_neg(&t); }
static g2j clear_cofactor(const g2j* p) {
    g2j t1 = g2_mul_x(p), t2 = g2_psi(p), d = g2_dbl(p), t3 = g2_psi(&d); t3 = g2_psi(&t3);
    g2j n = g2_neg(&t2); t3 = g2_add(&t3, &n); t2 = g2_add(&t1, &t2); t2 = g2_mul_x(&t2); t3 = g2_add(&t3, &t2);
    n = g2_neg(&t1); t3 = g2_add(&t3, &n); n = g2_neg(p); return g2_add(&t3, &n);
}
static g2a hash_to_g2(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen) {
    uint8_t uni[256]; expand_xmd(msg, mlen, dst, dlen, uni, 256);
    fp2 u0, u1; u0.c0 = fp_from_be64(uni); u0.c1 = fp_from_be64(uni + 64); u1.c0 = fp_from_be64(uni + 128); u1.c1 = fp_from_be64(uni + 192);
    g2a q0 = sswu(&u0), q1 = sswu(&u1); q0 = iso3(&q0); q1 = iso3(&q1);
    g2j a = g2_from_aff(&q0), b = g2_from_aff(&q1), s = g2_add(&a, &b), c = clear_cofactor(&s);
    return g2_to_aff(&c);
}

/* ------------------------------------------------------------------ pairing */
/* Miller loop over up to 8 affine pairs with shared squarings (blst_miller_loop_n shape) */
static fp12 miller_n(const g2a* Q, const g1a* P, int n) {
    g2j T[8]; fp12 f = f12_one();
    for (int i = 0; i < n; i++) T[i] = g2_from_aff(&Q[i]);
    for (int bit = 62; bit >= 0; bit--) {
        f = f12_sqr(&f);
        for (int i = 0; i < n; i++) {
            if (Q[i].inf || P[i].inf) continue;
            g2j* t = &T[i];
            fp2 A = f2_sqr(&t->x), B = f2_sqr(&t->y), zz = f2_sqr(&t->z), E = f2_add(&A, &A); E = f2_add(&E, &A);
            g2j d = g2_dbl(t);
            fp2 l0 = f2_mul(&E, &t->x), b2 = f2_dbl(&B); l0 = f2_sub(&l0, &b2);
            fp2 l1 = f2_mul(&E, &zz); l1 = f2_mul_fp(&l1, &P[i].x); l1 = f2_neg(&l1);
            fp2 l2 = f2_mul(&d.z, &zz); l2 = f2_mul_fp(&l2, &P[i].y);
            f = f12_mul_line(&f, &l0, &l1, &l2); *t = d;
        }
        if ((X_ABS >> bit) & 1)
            for (int i = 0; i < n; i++) {
                if (Q[i].inf || P[i].inf) continue;
                g2j* t = &T[i]; g2j qj = g2_from_aff(&Q[i]);
                /* chord through T and Q: slope rr / z3 with rr = yq*z^3 - y, h = xq*z^2 - x, z3 = z*h */
                fp2 zz = f2_sqr(&t->z), zzz = f2_mul(&zz, &t->z), u2 = f2_mul(&Q[i].x, &zz), s2 = f2_mul(&Q[i].y, &zzz);
                fp2 h = f2_sub(&u2, &t->x), rr = f2_sub(&s2, &t->y), z3 = f2_mul(&t->z, &h);
                fp2 l0 = f2_mul(&rr, &Q[i].x), yz = f2_mul(&Q[i].y, &z3); l0 = f2_sub(&l0, &yz);
                fp2 l1 = f2_mul_fp(&rr, &P[i].x); l1 = f2_neg(&l1);
                fp2 l2 = f2_mul_fp(&z3, &P[i].y);
                f = f12_mul_line(&f, &l0, &l1, &l2); *t = g2_add(t, &qj);
            }
    }
    return f12_conj(&f);
}
static fp12 cyc_exp_x(const fp12* a) {
    fp12 r = *a;
    for (int bit = 62; bit >= 0; bit--) { r = f12_sqr(&r); if ((X_ABS >> bit) & 1) r = f12_mul(&r, a); }
    return f12_conj(&r);
}
static fp12 final_exp(const fp12* f) {   /* f^(3 (p^12-1)/r) */
    fp12 fi = f12_inv(f), t = f12_conj(f); t = f12_mul(&t, &fi);
    fp12 t2 = f12_frob(&t); t2 = f12_frob(&t2); t = f12_mul(&t2, &t);
    fp12 tc = f12_conj(&t), a = cyc_exp_x(&t); a = f12_mul(&a, &tc);
    fp12 ac = f12_conj(&a), a2 = cyc_exp_x(&a); a = f12_mul(&a2, &ac);
    fp12 b = cyc_exp_x(&a), af = f12_frob(&a); b = f12_mul(&b, &af);
    fp12 c = cyc_exp_x(&b); c = cyc_exp_x(&c);
    fp12 bf = f12_frob(&b); bf = f12_frob(&bf); c = f12_mul(&c, &bf); fp12 bc = f12_conj(&b); c = f12_mul(&c, &bc);
    fp12 t3 = f12_sqr(&t); t3 = f12_mul(&t3, &t);
    return f12_mul(&c, &t3);
}

/* ------------------------------------------------------------------ pairing context (blst_pairing restated) */
typedef struct {
    fp12 gt; int gt_set; g2j aggr; int nq; g2a Q[8]; g1a P[8];
    uint8_t seed[32]; const uint8_t* dst; size_t dlen;
} pctx;
static void ctx_init(pctx* c, const uint8_t rnd[32], const uint8_t* tag, size_t taglen, const uint8_t* dst, size_t dlen) {
    c->gt = f12_one(); c->gt_set = 0; c->aggr = g2_inf(); c->nq = 0; c->dst = dst; c->dlen = dlen;
    sha_t s; sha_init(&s); sha_update(&s, rnd, 32); if (taglen) sha_update(&s, tag, taglen); sha_final(&s, c->seed);
}
static void ctx_flush(pctx* c) {
    if (c->nq == 0) return;
    fp12 m = miller_n(c->Q, c->P, c->nq);
    c->gt = c->gt_set ? f12_mul(&c->gt, &m) : m; c->gt_set = 1; c->nq = 0;
}
static uint64_t ctx_next_scalar(pctx* c) {
    uint64_t r;
    do { oracle_sha256(c->seed, 32, c->seed); r = 0; for (int i = 0; i < 8; i++) r |= (uint64_t)c->seed[i] << (8 * i); } while (r == 0);
    return r;
}
/* update(): returns 0 on BLST_PK_IS_INFINITY */
static int ctx_update(pctx* c, const uint8_t* set320, uint64_t* r_out, uint8_t* h_out, uint8_t* rpk_out) {
    uint64_t r = ctx_next_scalar(c);
    if (r_out) *r_out = r;
    g1a pk = ld_g1a(set320); g2a sig = ld_g2a(set320 + 128);
    if (pk.inf) return 0;
    if (!sig.inf) { g2j sj = g2_from_aff(&sig), rs = g2_mul_u64(&sj, r); c->aggr = g2_add(&c->aggr, &rs); }
    g2a h = hash_to_g2(set320 + 96, 32, c->dst, c->dlen);
    g1j pj = g1_from_aff(&pk), rp = g1_mul_u64(&pj, r); g1a rpa = g1_to_aff(&rp);
    if (h_out) st_g2a(h_out, &h);
    if (rpk_out) st_g1a(rpk_out, &rpa);
    c->Q[c->nq] = h; c->P[c->nq] = rpa; c->nq++;
    if (c->nq == 8) ctx_flush(c);
    return 1;
}
static void ctx_merge(pctx* a, const pctx* b) {
    if (b->gt_set) { a->gt = a->gt_set ? f12_mul(&a->gt, &b->gt) : b->gt; a->gt_set = 1; }
    a->aggr = g2_add(&a->aggr, &b->aggr);
}
static int ctx_finalverify(pctx* c, uint8_t* gt_out, uint8_t* agg_out) {
    g2a s = g2_to_aff(&c->aggr);
    if (agg_out) st_g2a(agg_out, &s);
    fp12 f = c->gt_set ? c->gt : f12_one();
    if (!s.inf) {
        g1a ng; ng.x = fp_c(K_G1X); fp gy = fp_c(K_G1Y); ng.y = fp_neg(&gy); ng.inf = 0;
        fp12 m = miller_n(&s, &ng, 1); f = f12_mul(&f, &m);
    }
    fp12 e = final_exp(&f);
    if (gt_out) st_f12(gt_out, &e);
    return f12_is_one(&e);
}

static const uint8_t DST_SIG[] = "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";

/* batchVerifySerial (nthreads == 0) / batchVerifyParallel with B = min(n, nthreads) contexts.
 * Optional stage outputs (may be NULL): r (n u64), H (n x 192 affine), rPK (n x 96), aggsig (192), gt (576). */
int oracle_batch_verify(const uint8_t* sets, size_t n, const uint8_t rnd[32], int nthreads,
                        uint64_t* r_out, uint8_t* h_out, uint8_t* rpk_out, uint8_t* agg_out, uint8_t* gt_out) {
    if (n == 0) return 0;
    size_t B = nthreads <= 0 ? 1 : ((size_t)nthreads < n ? (size_t)nthreads : n);
    pctx* ctx = (pctx*)malloc(B * sizeof(pctx));
    int* ok = (int*)malloc(B * sizeof(int));
    size_t base = n / B, rem = n % B;
#pragma omp parallel for schedule(static, 1)
    for (long c = 0; c < (long)B; c++) {
        size_t off = (size_t)c < rem ? (base + 1) * c : base * c + rem, len = (size_t)c < rem ? base + 1 : base;
        uint8_t tag[8]; u64_le(tag, (uint64_t)c);
        ctx_init(&ctx[c], rnd, tag, nthreads <= 0 ? 0 : 8, DST_SIG, sizeof(DST_SIG) - 1);
        ok[c] = 1;
        for (size_t i = off; i < off + len; i++)
            if (!ctx_update(&ctx[c], sets + 320 * i, r_out ? r_out + i : NULL, h_out ? h_out + 192 * i : NULL, rpk_out ? rpk_out + 96 * i : NULL)) { ok[c] = 0; break; }
        ctx_flush(&ctx[c]);   /* commit */
    }
    int all = 1;
    for (size_t c = 0; c < B; c++) all &= ok[c];
    int res = 0;
    if (all) { for (size_t c = 1; c < B; c++) ctx_merge(&ctx[0], &ctx[c]); res = ctx_finalverify(&ctx[0], gt_out, agg_out); }
    free(ctx); free(ok);
    return res;
}

/* aggregateVerify (bls_sig_min_pubkey.nim:153-199 over ContextCoreAggregateVerify, blst_min_pubkey_sig_core.nim:305-414): update per
 * (public key, message) pair WITHOUT blinding (chk_n_aggr_pk_in_g1: the pair (pk_i, H(m_i)) enters the Miller loop as it is), then
 * finish(signature): the signature is aggregated, commit, finalverify.  Messages: msgs[offs[i] .. offs[i+1]).  Pairs are split over
 * OpenMP threads, one pairing context each, merged in order (the GT product is commutative).  gt_out (optional): 576 B. */
int oracle_aggregate_verify(const uint8_t* pks, const uint8_t* msgs, const uint32_t* offs, size_t n, const uint8_t sig192[192], uint8_t* gt_out) {
    if (n == 0) return 0;
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
#endif
    if ((size_t)T > n) T = (int)n;
    pctx* ctx = (pctx*)malloc((size_t)T * sizeof(pctx));
    int* ok = (int*)malloc((size_t)T * sizeof(int));
    uint8_t zero[32] = {0};
    size_t base = n / T, rem = n % T;
#pragma omp parallel for schedule(static, 1)
    for (long c = 0; c < (long)T; c++) {
        size_t off = (size_t)c < rem ? (base + 1) * c : base * c + rem, len = (size_t)c < rem ? base + 1 : base;
        ctx_init(&ctx[c], zero, NULL, 0, DST_SIG, sizeof(DST_SIG) - 1);
        ok[c] = 1;
        for (size_t i = off; i < off + len; i++) {
            g1a pk = ld_g1a(pks + 96 * i);
            if (pk.inf) { ok[c] = 0; break; }                     /* BLST_PK_IS_INFINITY -> update false */
            g2a h = hash_to_g2(msgs + offs[i], offs[i + 1] - offs[i], DST_SIG, sizeof(DST_SIG) - 1);
            ctx[c].Q[ctx[c].nq] = h; ctx[c].P[ctx[c].nq] = pk; ctx[c].nq++;
            if (ctx[c].nq == 8) ctx_flush(&ctx[c]);
        }
        ctx_flush(&ctx[c]);
    }
    int all = 1;
    for (int c = 0; c < T; c++) all &= ok[c];
    int res = 0;
    if (all) {
        for (int c = 1; c < T; c++) ctx_merge(&ctx[0], &ctx[c]);
        g2a sg = ld_g2a(sig192);
        if (!sg.inf) { g2j sj = g2_from_aff(&sg); ctx[0].aggr = g2_add(&ctx[0].aggr, &sj); }
        res = ctx_finalverify(&ctx[0], gt_out, NULL);
    }
    free(ctx); free(ok);
    return res;
}
/* sum of n affine G2 points (aggregate signature of test inputs) */
void oracle_g2_sum(const uint8_t* pts, size_t n, uint8_t out192[192]) {
    g2j acc = g2_inf();
    for (size_t i = 0; i < n; i++) { g2a a = ld_g2a(pts + 192 * i); if (a.inf) continue; g2j j = g2_from_aff(&a); acc = g2_add(&acc, &j); }
    g2a r = g2_to_aff(&acc); st_g2a(out192, &r);
}

void oracle_hash_to_g2(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t out192[192]) {
    g2a h = hash_to_g2(msg, mlen, dst, dlen); st_g2a(out192, &h);
}
/* sk: 32-byte little-endian scalar */
void oracle_sk_to_pk(const uint8_t sk_le[32], uint8_t out96[96]) {
    g1j g; g.x = fp_c(K_G1X); g.y = fp_c(K_G1Y); g.z = fp_one(); g1j p = g1_mul(&g, sk_le, 256); g1a a = g1_to_aff(&p); st_g1a(out96, &a);
}
void oracle_sign(const uint8_t sk_le[32], const uint8_t* msg, size_t mlen, uint8_t out192[192]) {
    g2a h = hash_to_g2(msg, mlen, DST_SIG, sizeof(DST_SIG) - 1); g2j hj = g2_from_aff(&h), s = g2_mul(&hj, sk_le, 256); g2a a = g2_to_aff(&s); st_g2a(out192, &a);
}
/* [k]H for a precomputed affine G2 point (same-message signing shortcut) */
void oracle_g2_mul(const uint8_t in192[192], const uint8_t k_le[32], uint8_t out192[192]) {
    g2a h = ld_g2a(in192); g2j hj = g2_from_aff(&h), s = g2_mul(&hj, k_le, 256); g2a a = g2_to_aff(&s); st_g2a(out192, &a);
}
/* synthetic batch: tuple i = (pk_i, SHA256("msg" || decimal(i)), sig_i), sk_i = SHA256("sk" || LE64(seed+i)) with top bits cleared */
void oracle_make_batch(uint8_t* sets, size_t n, uint64_t seed) {
#pragma omp parallel for schedule(dynamic, 4)
    for (long i = 0; i < (long)n; i++) {
        uint8_t sk[32], in[10] = {'s', 'k'}, msg[32]; char txt[32];
        u64_le(in + 2, seed + (uint64_t)i); oracle_sha256(in, 10, sk); sk[31] &= 0x3f; sk[0] |= 1;
        int l = 0; { unsigned long long v = (unsigned long long)i; char tmp[24]; int k = 0; do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v); txt[0] = 'm'; txt[1] = 's'; txt[2] = 'g'; l = 3; while (k) txt[l++] = tmp[--k]; }
        oracle_sha256((const uint8_t*)txt, (size_t)l, msg);
        uint8_t* o = sets + 320 * (size_t)i;
        oracle_sk_to_pk(sk, o); memcpy(o + 96, msg, 32); oracle_sign(sk, msg, 32, o + 128);
    }
}
/* public keys of the same secret keys oracle_make_batch derives: sk_i = SHA256("sk" || LE64(seed+i)), top 2 bits cleared, bit 0 set */
void oracle_make_pks(uint8_t* pks, size_t n, uint64_t seed) {
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {
        uint8_t sk[32], in[10] = {'s', 'k'};
        u64_le(in + 2, seed + (uint64_t)i); oracle_sha256(in, 10, sk); sk[31] &= 0x3f; sk[0] |= 1;
        oracle_sk_to_pk(sk, pks + 96 * (size_t)i);
    }
}
/* ------------------------------------------------------------------ deserialisation (bls_sig_io.nim:42-99), by the definitions:
 * decompression through a square root, subgroup membership through [r]P == infinity */
static int fp_from_be48(fp* out, const uint8_t* b, int clear_flags) {
    fp v;
    for (int i = 0; i < 6; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[40 - 8 * i + j]; v.l[i] = w; }
    if (clear_flags) v.l[5] &= 0x1fffffffffffffffull;
    u128 br = 0;
    for (int i = 0; i < 6; i++) { u128 d = (u128)v.l[i] - K_P[i] - (uint64_t)br; br = (d >> 64) & 1; }
    if (!br) return 0;                       /* >= p */
    *out = fp_to_mont(&v); return 1;
}
static int fp_is_large(const fp* y_mont) {  /* canonical y > (p-1)/2 */
    fp y = fp_from_mont(y_mont);
    for (int i = 5; i >= 0; i--) { if (y.l[i] > K_PM1D2_INT[i]) return 1; if (y.l[i] < K_PM1D2_INT[i]) return 0; }
    return 0;
}
/* returns 0 ok, 1 bad encoding; *inf for the infinity encoding */
static int g1_uncompress(g1a* out, const uint8_t* b) {
    memset(out, 0, sizeof *out);
    if (!(b[0] & 0x80)) return 1;
    if (b[0] & 0x40) { int any = b[0] & 0x3f; for (int i = 1; i < 48; i++) any |= b[i]; out->inf = 1; return any != 0; }
    fp x; if (!fp_from_be48(&x, b, 1)) return 1;
    fp x2 = fp_sqr(&x), x3 = fp_mul(&x2, &x), four = fp_small(4), rhs = fp_add(&x3, &four), y;
    if (!fp_sqrt(&y, &rhs)) return 1;
    if (fp_is_large(&y) != ((b[0] >> 5) & 1)) y = fp_neg(&y);
    out->x = x; out->y = y; return 0;
}
static int g2_uncompress(g2a* out, const uint8_t* b) {
    memset(out, 0, sizeof *out);
    if (!(b[0] & 0x80)) return 1;
    if (b[0] & 0x40) { int any = b[0] & 0x3f; for (int i = 1; i < 96; i++) any |= b[i]; out->inf = 1; return any != 0; }
    fp2 x; if (!fp_from_be48(&x.c1, b, 1) || !fp_from_be48(&x.c0, b + 48, 0)) return 1;
    fp2 x2 = f2_sqr(&x), x3 = f2_mul(&x2, &x), b2; b2.c0 = fp_small(4); b2.c1 = b2.c0;
    fp2 rhs = f2_add(&x3, &b2), y;
    if (!f2_sqrt(&y, &rhs)) return 1;
    int large = fp_is_zero(&y.c1) ? fp_is_large(&y.c0) : fp_is_large(&y.c1);
    if (large != ((b[0] >> 5) & 1)) y = f2_neg(&y);
    out->x = x; out->y = y; return 0;
}
/* blst_p1_deserialize / blst_p2_deserialize semantics for the 96- / 192-byte forms: 000 uncompressed (big-endian coordinates
 * < p, on the curve), 1xx compressed, 01x infinity (0x40 then zeros), else bad.  0 ok, 1 bad. */
static int g1_deserialize(g1a* out, const uint8_t* b) {
    if (b[0] & 0x80) return g1_uncompress(out, b);
    memset(out, 0, sizeof *out);
    if (b[0] & 0x40) { int any = b[0] & 0x3f; for (int i = 1; i < 96; i++) any |= b[i]; out->inf = 1; return any != 0; }
    if (b[0] & 0x20) return 1;
    fp x, y; if (!fp_from_be48(&x, b, 0) || !fp_from_be48(&y, b + 48, 0)) return 1;
    fp x2 = fp_sqr(&x), x3 = fp_mul(&x2, &x), four = fp_small(4), rhs = fp_add(&x3, &four), y2 = fp_sqr(&y);
    if (!fp_eq(&y2, &rhs)) return 1;
    out->x = x; out->y = y; return 0;
}
static int g2_deserialize(g2a* out, const uint8_t* b) {
    if (b[0] & 0x80) return g2_uncompress(out, b);
    memset(out, 0, sizeof *out);
    if (b[0] & 0x40) { int any = b[0] & 0x3f; for (int i = 1; i < 192; i++) any |= b[i]; out->inf = 1; return any != 0; }
    if (b[0] & 0x20) return 1;
    fp2 x, y;
    if (!fp_from_be48(&x.c1, b, 0) || !fp_from_be48(&x.c0, b + 48, 0) || !fp_from_be48(&y.c1, b + 96, 0) || !fp_from_be48(&y.c0, b + 144, 0)) return 1;
    fp2 x2 = f2_sqr(&x), x3 = f2_mul(&x2, &x), b2; b2.c0 = fp_small(4); b2.c1 = fp_small(4);
    fp2 rhs = f2_add(&x3, &b2), y2 = f2_sqr(&y);
    if (!f2_eq(&y2, &rhs)) return 1;
    out->x = x; out->y = y; return 0;
}
/* status bytes and flags as in include/blscurve_mi355x.h (1: 96-byte keys, 2: 192-byte signatures, 4: KnownOnCurve = no subgroup
 * checks); out320 may be NULL.  Returns 1 when every tuple deserialised. */
int oracle_deserialize_sets_ex(const uint8_t* pks, const uint8_t* msgs, const uint8_t* sigs, size_t n, unsigned flags, uint8_t* out320, uint8_t* status) {
    int all = 1;
    size_t pkb = (flags & 1) ? 96 : 48, sgb = (flags & 2) ? 192 : 96;
#pragma omp parallel for schedule(dynamic, 8) reduction(& : all)
    for (long i = 0; i < (long)n; i++) {
        g1a pk; g2a sg; uint8_t st = 0;
        if ((flags & 1) ? g1_deserialize(&pk, pks + pkb * i) : g1_uncompress(&pk, pks + pkb * i)) st = 1;
        else if (pk.inf) st = 3;
        else if (!(flags & 4)) { g1j pj = g1_from_aff(&pk), t = g1_mul(&pj, K_R_LE, 255); if (!g1_is_inf(&t)) st = 2; }
        if (!st) {
            if ((flags & 2) ? g2_deserialize(&sg, sigs + sgb * i) : g2_uncompress(&sg, sigs + sgb * i)) st = 4;
            else if (!sg.inf && !(flags & 4)) { g2j sj = g2_from_aff(&sg), t = g2_mul(&sj, K_R_LE, 255); if (!g2_is_inf(&t)) st = 5; }
        }
        if (status) status[i] = st;
        if (st) all = 0;
        if (out320) {
            uint8_t* o = out320 + 320 * (size_t)i;
            if (st) memset(o, 0, 320);
            else { st_g1a(o, &pk); memcpy(o + 96, msgs + 32 * i, 32); st_g2a(o + 128, &sg); }
            if (st) memcpy(o + 96, msgs + 32 * i, 32);
        }
    }
    return all;
}
int oracle_deserialize_sets(const uint8_t* pks, const uint8_t* msgs, const uint8_t* sigs, size_t n, uint8_t* out320, uint8_t* status) {
    return oracle_deserialize_sets_ex(pks, msgs, sigs, n, 0, out320, status);
}
/* uncompressed serialisation of the 320-byte records (96-byte keys, 192-byte signatures; infinity = 0x40 then zeros) */
void oracle_serialize_sets(const uint8_t* sets, size_t n, uint8_t* pks96, uint8_t* sigs192) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t* r = sets + 320 * i;
        g1a pk = ld_g1a(r); g2a sg = ld_g2a(r + 128);
        uint8_t* o = pks96 + 96 * i; memset(o, 0, 96);
        if (pk.inf) o[0] = 0x40;
        else { fp c[2] = {fp_from_mont(&pk.x), fp_from_mont(&pk.y)}; for (int q = 0; q < 2; q++) for (int k = 0; k < 6; k++) for (int j = 0; j < 8; j++) o[48 * q + 40 - 8 * k + j] = (uint8_t)(c[q].l[k] >> (56 - 8 * j)); }
        o = sigs192 + 192 * i; memset(o, 0, 192);
        if (sg.inf) o[0] = 0x40;
        else { fp c[4] = {fp_from_mont(&sg.x.c1), fp_from_mont(&sg.x.c0), fp_from_mont(&sg.y.c1), fp_from_mont(&sg.y.c0)}; for (int q = 0; q < 4; q++) for (int k = 0; k < 6; k++) for (int j = 0; j < 8; j++) o[48 * q + 40 - 8 * k + j] = (uint8_t)(c[q].l[k] >> (56 - 8 * j)); }
    }
}
/* ZCash compression of the 320-byte records (for building wire-format test inputs) */
void oracle_compress_sets(const uint8_t* sets, size_t n, uint8_t* pks48, uint8_t* msgs32, uint8_t* sigs96) {
    for (size_t i = 0; i < n; i++) {
        const uint8_t* r = sets + 320 * i;
        g1a pk = ld_g1a(r); g2a sg = ld_g2a(r + 128);
        uint8_t* o = pks48 + 48 * i; memset(o, 0, 48);
        if (pk.inf) o[0] = 0xc0;
        else { fp x = fp_from_mont(&pk.x); for (int k = 0; k < 6; k++) for (int j = 0; j < 8; j++) o[40 - 8 * k + j] = (uint8_t)(x.l[k] >> (56 - 8 * j)); o[0] |= 0x80 | (fp_is_large(&pk.y) ? 0x20 : 0); }
        memcpy(msgs32 + 32 * i, r + 96, 32);
        o = sigs96 + 96 * i; memset(o, 0, 96);
        if (sg.inf) o[0] = 0xc0;
        else {
            fp x1 = fp_from_mont(&sg.x.c1), x0 = fp_from_mont(&sg.x.c0);
            for (int k = 0; k < 6; k++) for (int j = 0; j < 8; j++) { o[40 - 8 * k + j] = (uint8_t)(x1.l[k] >> (56 - 8 * j)); o[48 + 40 - 8 * k + j] = (uint8_t)(x0.l[k] >> (56 - 8 * j)); }
            int large = fp_is_zero(&sg.y.c1) ? fp_is_large(&sg.y.c0) : fp_is_large(&sg.y.c1);
            o[0] |= 0x80 | (large ? 0x20 : 0);
        }
    }
}

/* sum of affine G1 points (aggregateAll, core :179-195) -> affine */
void oracle_g1_sum(const uint8_t* pts, size_t n, uint8_t out96[96]) {
    g1j acc = g1_inf();
    for (size_t i = 0; i < n; i++) { g1a a = ld_g1a(pts + 96 * i); g1j j = g1_from_aff(&a); acc = g1_add(&acc, &j); }
    g1a r = g1_to_aff(&acc); st_g1a(out96, &r);
}
/* fastAggregateVerify (bls_sig_min_pubkey.nim:234-258) */
int oracle_fast_aggregate_verify(const uint8_t* pks, size_t n, const uint8_t* msg, size_t mlen, const uint8_t sig192[192]) {
    if (n == 0) return 0;
    uint8_t agg[96]; oracle_g1_sum(pks, n, agg);
    g1a pk = ld_g1a(agg); if (pk.inf) return 0;
    g2a Q[2]; g1a P[2];
    Q[0] = hash_to_g2(msg, mlen, DST_SIG, sizeof(DST_SIG) - 1); P[0] = pk;
    Q[1] = ld_g2a(sig192); P[1].x = fp_c(K_G1X); fp gy = fp_c(K_G1Y); P[1].y = fp_neg(&gy); P[1].inf = 0;
    fp12 f = miller_n(Q, P, 2), e = final_exp(&f);
    return f12_is_one(&e);
}
/* naive MSM: sum [k_i mod 2^nbits] P_i, scalars 32-byte LE (blst_p1s_mult_pippenger semantics) -> affine */
void oracle_msm_g1(const uint8_t* pts, const uint8_t* scalars, size_t n, int nbits, uint8_t out96[96]) {
    g1j acc = g1_inf();
#pragma omp parallel
    {
        g1j loc = g1_inf();
#pragma omp for schedule(static)
        for (long i = 0; i < (long)n; i++) { g1a a = ld_g1a(pts + 96 * i); g1j j = g1_from_aff(&a), m = g1_mul(&j, scalars + 32 * i, nbits); loc = g1_add(&loc, &m); }
#pragma omp critical
        acc = g1_add(&acc, &loc);
    }
    g1a r = g1_to_aff(&acc); st_g1a(out96, &r);
}
/* Pippenger bucket method (the published algorithm blst_p1s_mult_pippenger implements; restated, not copied: BLST's source is
 * absent): the nbits are cut into windows of c bits; per window every point is added into the bucket of its digit, the buckets are
 * integrated with a running sum (sum_d d * B_d), and the window results are combined by c doublings each.  Windows run on
 * OpenMP threads.  Same semantics as oracle_msm_g1 (scalars taken mod 2^nbits, 32-byte little-endian). */
static uint32_t msm_digit(const uint8_t* k, int bit0, int c, int nbits) {
    uint32_t d = 0;
    for (int j = 0; j < c && bit0 + j < nbits; j++) d |= (uint32_t)((k[(bit0 + j) >> 3] >> ((bit0 + j) & 7)) & 1) << j;
    return d;
}
void oracle_msm_g1_pippenger(const uint8_t* pts, const uint8_t* scalars, size_t n, int nbits, uint8_t out96[96]) {
    int c = 1;
    while (c < 16 && ((size_t)1 << (c + 3)) <= n) c++;          /* about log2(n) - 2 */
    int nwin = (nbits + c - 1) / c;
    g1j* win = (g1j*)malloc((size_t)nwin * sizeof(g1j));
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < nwin; w++) {
        size_t nb = (size_t)1 << c;
        g1j* B = (g1j*)malloc(nb * sizeof(g1j));
        for (size_t b = 0; b < nb; b++) B[b] = g1_inf();
        for (size_t i = 0; i < n; i++) {
            uint32_t d = msm_digit(scalars + 32 * i, w * c, c, nbits);
            if (!d) continue;
            g1a a = ld_g1a(pts + 96 * i); g1j j = g1_from_aff(&a);
            B[d] = g1_add(&B[d], &j);
        }
        g1j run = g1_inf(), acc = g1_inf();
        for (size_t b = nb - 1; b >= 1; b--) { run = g1_add(&run, &B[b]); acc = g1_add(&acc, &run); }
        win[w] = acc;
        free(B);
    }
    g1j acc = g1_inf();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int j = 0; j < c; j++) acc = g1_dbl(&acc);
        acc = g1_add(&acc, &win[w]);
    }
    free(win);
    g1a r = g1_to_aff(&acc); st_g1a(out96, &r);
}
/* naive G2 MSM (blst_p2s_mult_pippenger semantics): sum [k_i mod 2^nbits] Q_i, scalars `sbytes` apart -> affine */
void oracle_msm_g2(const uint8_t* pts, const uint8_t* scalars, size_t n, int sbytes, int nbits, uint8_t out192[192]) {
    g2j acc = g2_inf();
#pragma omp parallel
    {
        g2j loc = g2_inf();
#pragma omp for schedule(static)
        for (long i = 0; i < (long)n; i++) { g2a a = ld_g2a(pts + 192 * i); g2j j = g2_from_aff(&a), m = g2_mul(&j, scalars + (size_t)sbytes * i, nbits); loc = g2_add(&loc, &m); }
#pragma omp critical
        acc = g2_add(&acc, &loc);
    }
    g2a r = g2_to_aff(&acc); st_g2a(out192, &r);
}
/* The same bucket method on G2 (blst_p2s_mult_pippenger, blst_abi.nim:358-362; call site blst_min_pubkey_sig_core.nim:639-646), scalars
 * `sbytes` apart: the checker of the device G2 MSM at sizes the naive loop above does not finish in seconds. */
static uint32_t msm_digit_n(const uint8_t* k, int bit0, int c, int nbits, int sbytes) {
    uint32_t d = 0;
    for (int j = 0; j < c && bit0 + j < nbits && ((bit0 + j) >> 3) < sbytes; j++) d |= (uint32_t)((k[(bit0 + j) >> 3] >> ((bit0 + j) & 7)) & 1) << j;
    return d;
}
void oracle_msm_g2_pippenger(const uint8_t* pts, const uint8_t* scalars, size_t n, int sbytes, int nbits, uint8_t out192[192]) {
    int c = 1;
    while (c < 14 && ((size_t)1 << (c + 3)) <= n) c++;
    int nwin = (nbits + c - 1) / c;
    g2j* win = (g2j*)malloc((size_t)nwin * sizeof(g2j));
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < nwin; w++) {
        size_t nb = (size_t)1 << c;
        g2j* B = (g2j*)malloc(nb * sizeof(g2j));
        for (size_t b = 0; b < nb; b++) B[b] = g2_inf();
        for (size_t i = 0; i < n; i++) {
            uint32_t d = msm_digit_n(scalars + (size_t)sbytes * i, w * c, c, nbits, sbytes);
            if (!d) continue;
            g2a a = ld_g2a(pts + 192 * i); g2j j = g2_from_aff(&a);
            B[d] = g2_add(&B[d], &j);
        }
        g2j run = g2_inf(), acc = g2_inf();
        for (size_t b = nb - 1; b >= 1; b--) { run = g2_add(&run, &B[b]); acc = g2_add(&acc, &run); }
        win[w] = acc;
        free(B);
    }
    g2j acc = g2_inf();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int j = 0; j < c; j++) acc = g2_dbl(&acc);
        acc = g2_add(&acc, &win[w]);
    }
    free(win);
    g2a r = g2_to_aff(&acc); st_g2a(out192, &r);
}
/* combine (blst_min_pubkey_sig_core.nim:570-647): scalars = u64 words 3,2,1,0 of a SHA-256 chain seeded with rnd itself, zeros
 * skipped; out_pk = sum [s_i]PK_i, out_sig = sum [s_i]S_i, affine.  scalars_out (optional): n u64. */
void oracle_combine(const uint8_t rnd[32], const uint8_t* pks, const uint8_t* sigs, size_t n, uint8_t out_pk[96], uint8_t out_sig[192], uint64_t* scalars_out) {
    uint64_t* sc = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint8_t seed[32]; memcpy(seed, rnd, 32);
    int avail = 0;
    for (size_t i = 0; i < n; i++) {
        for (;;) {
            if (avail == 0) { uint8_t nx[32]; oracle_sha256(seed, 32, nx); memcpy(seed, nx, 32); avail = 4; }
            avail--;
            uint64_t w = 0; for (int j = 7; j >= 0; j--) w = (w << 8) | seed[8 * avail + j];
            if (w) { sc[i] = w; break; }
        }
    }
    g1j a1 = g1_inf(); g2j a2 = g2_inf();
#pragma omp parallel
    {
        g1j l1 = g1_inf(); g2j l2 = g2_inf();
#pragma omp for schedule(static)
        for (long i = 0; i < (long)n; i++) {
            g1a p = ld_g1a(pks + 96 * i); g1j pj = g1_from_aff(&p), m1 = g1_mul_u64(&pj, sc[i]); l1 = g1_add(&l1, &m1);
            g2a q = ld_g2a(sigs + 192 * i); g2j qj = g2_from_aff(&q), m2 = g2_mul_u64(&qj, sc[i]); l2 = g2_add(&l2, &m2);
        }
#pragma omp critical
        { a1 = g1_add(&a1, &l1); a2 = g2_add(&a2, &l2); }
    }
    g1a r1 = g1_to_aff(&a1); st_g1a(out_pk, &r1);
    g2a r2 = g2_to_aff(&a2); st_g2a(out_sig, &r2);
    if (scalars_out) memcpy(scalars_out, sc, n * sizeof(uint64_t));
    free(sc);
}
/* coreVerify (blst_min_pubkey_sig_core.nim:269-297; `verify`, bls_sig_min_pubkey.nim:100-107): e(pk, H(msg)) == e(G1, sig) */
int oracle_core_verify(const uint8_t pk96[96], const uint8_t* msg, size_t mlen, const uint8_t sig192[192]) {
    return oracle_fast_aggregate_verify(pk96, 1, msg, mlen, sig192);
}
int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
