// Lane-cooperative Fp12 multiplication / squaring for the per-batch SERIAL tail (Horner over the 68 step products, shard
// merge, final exponentiation): a handful of Fp12 operations in a dependent chain, so latency is all that matters.
// An Fp12 value is kept in the flat basis Fp2[w]/(w^6 - xi) (tower slots c0.(a0,a1,a2), c1.(a0,a1,a2) = w^0,2,4 / w^1,3,5).
// A product is two phases on a block of three waves, each a handful of instructions per lane (round 4; rounds 1-3 ran Karatsuba triples, a 168-item limb
// combination and twelve reducing lanes behind three barriers):
//   1   144 lanes (84 for a square): ONE Fp multiplication each - the four products x0 y0, x1 y1, x0 y1, x1 y0 of every coefficient pair
//   2   192 lanes = 12 coefficients x 16: lane (c, l) sums the 24 product limbs that feed limb l of coefficient c (exact in 64 bits), passes its carry to
//       the neighbour, takes the quotient of the partial reduction from lane 13 and stores its limb (|v| < 0.51 p, semi-normalised limbs)
// The pieces are __host__ __device__ and take the lane / item index as an argument: k_tail calls them with threadIdx.x
// between barriers, tests/host_emu runs the same code in a loop and compares with the tower's fp12_mul.
#pragma once
#include "tower.hpp"

namespace bls {

struct c12_work {
    fp prod[144];               // the four Fp products (x0 y0, x1 y1, x0 y1, x1 y0) of the 36 (or 21) coefficient pairs (the Karatsuba form: three, 108 slots)
    int32_t pl[16];             // the limbs of p (the row phase reads p_l by lane; filled once per kernel: c12_fill_p)
#ifdef BLS_TAIL_CLOCK
    unsigned long long prof[8]; // diagnostic build: time of thread 0 in each phase of the engine product
#endif
};

BLS_HD int c12_flat_of_tower(int t) { return t < 3 ? 2 * t : 2 * (t - 3) + 1; }
// index of the pair (i, j), i <= j, in the row-by-row enumeration used for squares: i = 0: j = 0..5 (6), i = 1: 5, ...
BLS_HD int c12_sqr_pair_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// ---- Schoolbook products: with one product per lane and lanes to spare, Karatsuba's saved product buys nothing, while its operand sums
// cost the critical wave 28 extra LDS reads and 28 additions in front of every multiplication.  Item q = 4 pr + kind of pair pr: kind 0: x0 y0, 1: x1 y1,
// 2: x0 y1, 3: x1 y0 (144 items for a product, 84 for a square) - every lane loads exactly two Fp operands, chosen by ADDRESS.
BLS_HD void c12s_item(int q, bool sqr, int& i, int& j, int& kx, int& ky) {
    const int pr = q >> 2, kind = q & 3;
    if (sqr) {                      // row-by-row enumeration of the pairs i <= j (rows start at 0, 6, 11, 15, 18, 20)
        i = (pr >= 6) + (pr >= 11) + (pr >= 15) + (pr >= 18) + (pr >= 20);
        j = i + pr - (i * 6 - (i * (i - 1)) / 2);
    } else {
        i = pr / 6;
        j = pr % 6;
    }
    kx = (kind == 1 || kind == 3) ? 1 : 0;
    ky = (kind == 1 || kind == 2) ? 1 : 0;
}
BLS_HD fp c12s_product(const fp2* A, const fp2* B, int q, bool sqr) {
    int i, j, kx, ky;
    c12s_item(q, sqr, i, j, kx, ky);
    const fp& x = kx ? A[i].c1 : A[i].c0;
    const fp& y = ky ? B[j].c1 : B[j].c0;
    return fp_mul(x, y);
}
// limb l of coefficient c (kk = c / 2, comp = c % 2) over the six terms: with t = (x0 y0, x1 y1, x0 y1, x1 y0) of a pair,
//   real:      t0 - t1           wrapped (times xi = 1 + u): t0 - t1 - (t2 + t3)
//   imaginary: t2 + t3           wrapped:                    t0 - t1 + (t2 + t3)
// A square's unordered pair {i, j}, i != j, is met twice (the same stored products): that is its factor two.  |sum| <= 24 * 2^28 per low limb.
BLS_HD int64_t c12s_limb_sum(const c12_work& W, int c, int l, bool sqr) {
    const int kk = c >> 1, comp = c & 1;
    int32_t v0[6], v1[6], v2[6], v3[6], cd[6], cx[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int j = kk - i;
        const bool wrap = j < 0;
        j += wrap ? 6 : 0;
        const int a = i < j ? i : j, b = i < j ? j : i;
        const int pr = sqr ? c12_sqr_pair_index(a, b) : i * 6 + j;
        cd[i] = comp ? (wrap ? 1 : 0) : 1;              // coefficient of (t0 - t1)
        cx[i] = comp ? 1 : (wrap ? -1 : 0);             // coefficient of (t2 + t3)
        const fp* t4 = &W.prod[4 * pr];
        v0[i] = (int32_t)t4[0].l[l];
        v1[i] = (int32_t)t4[1].l[l];
        v2[i] = (int32_t)t4[2].l[l];
        v3[i] = (int32_t)t4[3].l[l];
    }
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        s = bls_mac(s, cd[i], v0[i] - v1[i]);           // |difference| < 2^29, |sum| < 2^29: exact in 32 bits
        s = bls_mac(s, cx[i], v2[i] + v3[i]);
    }
    return s;
}

// ---- Phase 2 on rows (round 4): lane (c, l) = (t / 16, t % 16) of a 192-thread block keeps ITS limb in a register from the sum to the stored result;
// the two carry steps and the partial reduction of phases 2a / 2b (twelve lanes walking fourteen limbs each, behind a barrier) become per-lane
// arithmetic with the neighbour's carry (a DPP row shift on the device, an array in the host harness) and the quotient of lane 13.
// Pieces, each a pure function of one lane's values:
//   c12_split(s, l)          -> low 28 bits and carry of a signed 64-bit limb value (the top limb l = 13 keeps everything)
//   c12_quotient(top)        -> round(v / p) from the top limb + the carry-in it will receive, as fp_reduce does
//   c12_sub_qp(limb, q, l)   -> limb - q * p_l (64 bits)
// Result limbs lie in (-2^7, 2^28 + 2^7): semi-normalised (limb bound 1), |value| < 0.51 p like fp_reduce's.
struct c12_lc { int32_t lo, car; };
BLS_HD c12_lc c12_split(int64_t s, int l) {
    c12_lc r;
    if (l < FP_N - 1) { r.lo = (int32_t)(s & (int64_t)FP_MASK); r.car = (int32_t)(s >> 28); }
    else { r.lo = (int32_t)s; r.car = 0; }
    return r;
}
BLS_HD int32_t c12_quotient(int32_t top) {
    const int64_t RECIP = 10322735;                       // round(2^40 / (p / 2^364)), as in fp_reduce
    return (int32_t)(((int64_t)top * RECIP + (1ll << 39)) >> 40);
}
BLS_HD int32_t c12_p_limb(int l) {                      // limb l of p, by selection (l differs per lane)
    int32_t v = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) v = l == i ? (int32_t)k::P[i] : v;
    return v;
}
BLS_HD int64_t c12_sub_qp(int32_t limb, int32_t q, int l) { return (int64_t)limb - (int64_t)q * c12_p_limb(l); }
BLS_HD int64_t c12_sub_qp_v(int32_t limb, int32_t q, int32_t pl) { return (int64_t)limb - (int64_t)q * pl; }      // p_l from the table (14 compares and selects cost ~340 cycles per product)
BLS_HD void c12_fill_p(c12_work& W, int t) { if (t < 16) W.pl[t] = t < FP_N ? (int32_t)k::P[t] : 0; }
// the whole row step on arrays (host harness; the device composes the same pieces with DPP): s[16] -> out limbs [14]
BLS_HD void c12_row_reduce_ref(const int64_t (&s)[16], int32_t (&out)[FP_N]) {
    int32_t limb[16], car[16];
    for (int l = 0; l < 16; l++) { c12_lc x = c12_split(l < FP_N ? s[l] : 0, l < FP_N ? l : 0); limb[l] = x.lo; car[l] = l < FP_N ? x.car : 0; }
    for (int l = FP_N - 1; l > 0; l--) limb[l] += car[l - 1];
    const int32_t q = c12_quotient(limb[FP_N - 1] + (limb[FP_N - 2] >> 28));
    int32_t lo2[16], car2[16];
    for (int l = 0; l < FP_N; l++) { c12_lc x = c12_split(c12_sub_qp(limb[l], q, l), l); lo2[l] = x.lo; car2[l] = x.car; }
    for (int l = 0; l < FP_N; l++) out[l] = lo2[l] + (l ? car2[l - 1] : 0);
}

}  // namespace bls
