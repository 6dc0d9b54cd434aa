// Lane-cooperative Fp12 multiplication / squaring for the per-batch SERIAL tail (Horner over the 68 step products, shard
// merge, final exponentiation): a handful of Fp12 operations in a dependent chain, so latency is all that matters.
// An Fp12 value is kept in the flat basis Fp2[w]/(w^6 - xi) (tower slots c0.(a0,a1,a2), c1.(a0,a1,a2) = w^0,2,4 / w^1,3,5).
// A product is three phases, each a handful of instructions per lane:
//   1   108 lanes (63 for a square): ONE Fp multiplication each - the Karatsuba triple (t0, t1, s) of every coefficient pair
//   2a  168 (coefficient, limb) items: the signed limb-wise combination of the 18 products that feed the coefficient,
//       exact in 64 bits, split into a 28-bit limb and a carry for the next limb
//   2b  12 lanes: limb + carry, partial reduction (|v| < 0.51 p), result coefficient
// The pieces are __host__ __device__ and take the lane / item index as an argument: k_tail calls them with threadIdx.x
// between barriers, tests/host_emu runs the same code in a loop and compares with the tower's fp12_mul.
#pragma once
#include "tower.hpp"

namespace bls {

struct c12_work {
    fp prod[108];               // Karatsuba triples (t0, t1, s) of the 36 (or 21) coefficient pairs
    int32_t lo[12][FP_N];       // phase 2a: low 28 bits of every combined limb (top limb: the whole signed value)
    int32_t car[12][FP_N];      // phase 2a: carry into limb l (from limb l - 1)
};

BLS_HD int c12_flat_of_tower(int t) { return t < 3 ? 2 * t : 2 * (t - 3) + 1; }
// index of the pair (i, j), i <= j, in the row-by-row enumeration used for squares: i = 0: j = 0..5 (6), i = 1: 5, ...
BLS_HD int c12_sqr_pair_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// One Fp product of the Karatsuba triple of the coefficient pair (x, y): kind 0: x.c0*y.c0, 1: x.c1*y.c1,
// 2: (x.c0 + x.c1)(y.c0 + y.c1).
BLS_HD fp c12_triple(const fp2& x, const fp2& y, int kind) {
    fp u = kind == 0 ? x.c0 : (kind == 1 ? x.c1 : fp_add_nc(x.c0, x.c1));
    fp v = kind == 0 ? y.c0 : (kind == 1 ? y.c1 : fp_add_nc(y.c0, y.c1));
    return fp_mul(u, v);
}

// phase 1, item q (q < 108: product of A and B; sqr: q < 63, B == A)
BLS_HD fp c12_phase1(const fp2* A, const fp2* B, int q, bool sqr) {
    int pr = q / 3, kind = q % 3, i, j;
    if (sqr) {
        int base = 0;
        i = 0;
        while (pr >= base + (6 - i)) { base += 6 - i; i++; }
        j = i + (pr - base);
    } else {
        i = pr / 6;
        j = pr % 6;
    }
    return c12_triple(A[i], B[j], kind);
}

// phase 2a, item t < 168: coefficient c = t / 14 (kk = c / 2 the power of w, comp = c % 2 real / imaginary), limb l = t % 14.
// The Fp2 product of pair (i, j) lands on w^(i+j); wrapped terms (i + j >= 6) are multiplied by xi = 1 + u:
//   plain:   re = t0 - t1        im = s - t0 - t1
//   wrapped: re - im = 2 t0 - s  re + im = s - 2 t1
// Products have limbs 0..12 in [0, 2^28) and a small signed top limb, so the sum of at most 6 * 2 * 4 of them is exact in 64 bits.
BLS_HD void c12_phase2a(c12_work& W, int t, bool sqr) {
    int c = t / FP_N, l = t % FP_N, kk = c >> 1, comp = c & 1;
    int64_t s = 0;
    for (int i = 0; i < 6; i++) {
        int j = kk - i;
        bool wrap = j < 0;
        if (wrap) j += 6;
        int mult = 1, pr = i * 6 + j;
        if (sqr) {
            if (i > j) continue;                              // (j, i) is counted, doubled
            pr = c12_sqr_pair_index(i, j);
            mult = i == j ? 1 : 2;
        }
        int c0 = comp ? (wrap ? 0 : -1) : (wrap ? 2 : 1);
        int c1 = comp ? (wrap ? -2 : -1) : (wrap ? 0 : -1);
        int cs = comp ? 1 : (wrap ? -1 : 0);
        const fp* t3 = &W.prod[3 * pr];
        s += (int64_t)mult * (c0 * (int64_t)(int32_t)t3[0].l[l] + c1 * (int64_t)(int32_t)t3[1].l[l] + cs * (int64_t)(int32_t)t3[2].l[l]);
    }
    if (l < FP_N - 1) {
        W.lo[c][l] = (int32_t)(s & (int64_t)FP_MASK);
        W.car[c][l + 1] = (int32_t)(s >> 28);
    } else {
        W.lo[c][l] = (int32_t)s;
    }
    if (l == 0) W.car[c][0] = 0;
}

// phase 2b, coefficient c < 12: the value (|v| <= 36 p: six terms of at most three products of |v| < 2p each, doubled at most)
// partially reduced, so that bounds never accumulate along a chain of products
BLS_HD fp c12_phase2b(const c12_work& W, int c) {
    fp v;
#pragma unroll
    for (int l = 0; l < FP_N; l++) v.l[l] = (uint32_t)(W.lo[c][l] + W.car[c][l]);
    BLS_SET_VB(v, 36);
    BLS_SET_LB(v, 1);
    return fp_reduce(v);
}

}  // namespace bls
