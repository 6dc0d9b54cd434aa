// Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v), xi = 1+u.
// Memory order of fp12 = blst_fp12 (blst_abi.nim:100-108): c0.(a0,a1,a2), c1.(a0,a1,a2), each fp2.
#pragma once
#include "fp.hpp"

namespace bls {

struct fp6 {
    fp2 a0, a1, a2;
};
struct fp12 {
    fp6 c0, c1;
};

BLS_HD fp6 fp6_zero() { return fp6{fp2_zero(), fp2_zero(), fp2_zero()}; }
BLS_HD fp6 fp6_add(const fp6& a, const fp6& b) { return fp6{fp2_add(a.a0, b.a0), fp2_add(a.a1, b.a1), fp2_add(a.a2, b.a2)}; }
BLS_HD fp6 fp6_sub(const fp6& a, const fp6& b) { return fp6{fp2_sub(a.a0, b.a0), fp2_sub(a.a1, b.a1), fp2_sub(a.a2, b.a2)}; }
BLS_HD fp6 fp6_neg(const fp6& a) { return fp6{fp2_neg(a.a0), fp2_neg(a.a1), fp2_neg(a.a2)}; }
BLS_HD fp6 fp6_dbl(const fp6& a) { return fp6{fp2_dbl(a.a0), fp2_dbl(a.a1), fp2_dbl(a.a2)}; }
BLS_HD fp6 fp6_mul_by_v(const fp6& a) { return fp6{fp2_mul_xi(a.a2), a.a0, a.a1}; }
// partial reduction of every coefficient (|v| < 0.51p): Fp12-level results are stored reduced so that value
// bounds never accumulate across the Karatsuba layers
BLS_HD fp6 fp6_add_nc(const fp6& a, const fp6& b) { return fp6{fp2_add_nc(a.a0, b.a0), fp2_add_nc(a.a1, b.a1), fp2_add_nc(a.a2, b.a2)}; }
BLS_HD fp6 fp6_sub_nc(const fp6& a, const fp6& b) { return fp6{fp2_sub_nc(a.a0, b.a0), fp2_sub_nc(a.a1, b.a1), fp2_sub_nc(a.a2, b.a2)}; }
BLS_HD fp6 fp6_carry(const fp6& a) { return fp6{fp2_carry(a.a0), fp2_carry(a.a1), fp2_carry(a.a2)}; }
BLS_HD fp6 fp6_reduce(const fp6& a) { return fp6{fp2_reduce(a.a0), fp2_reduce(a.a1), fp2_reduce(a.a2)}; }

// 6 fp2 multiplications
// Karatsuba, 6 fp2 multiplications.  Operands carried (1 limb unit): their pairwise sums feed the multiplier
// un-carried; each result coefficient takes one carry step.
BLS_HDN fp6 fp6_mul_impl(const fp6& a, const fp6& b) {
    fp2 t0 = fp2_mul(a.a0, b.a0);
    fp2 t1 = fp2_mul(a.a1, b.a1);
    fp2 t2 = fp2_mul(a.a2, b.a2);
    fp2 c0 = fp2_sub_nc(fp2_sub_nc(fp2_mul(fp2_add_nc(a.a1, a.a2), fp2_add_nc(b.a1, b.a2)), t1), t2);       // 3 units
    c0 = fp2_carry(fp2_add_nc(t0, fp2_mul_xi_nc(c0)));                                                     // 1 + 6
    fp2 c1 = fp2_sub_nc(fp2_sub_nc(fp2_mul(fp2_add_nc(a.a0, a.a1), fp2_add_nc(b.a0, b.a1)), t0), t1);
    c1 = fp2_carry(fp2_add_nc(c1, fp2_mul_xi_nc(t2)));                                                     // 3 + 2
    fp2 c2 = fp2_sub_nc(fp2_sub_nc(fp2_mul(fp2_add_nc(a.a0, a.a2), fp2_add_nc(b.a0, b.a2)), t0), t2);
    c2 = fp2_carry(fp2_add_nc(c2, t1));
    return fp6{c0, c1, c2};
}

// by-value wrappers: see jac_add (curve.hpp)
BLS_HD fp6 fp6_mul(fp6 a, fp6 b) { return fp6_mul_impl(a, b); }

BLS_HDN fp6 fp6_inv(const fp6& a) {
    fp2 c0 = fp2_sub(fp2_sqr(a.a0), fp2_mul_xi(fp2_mul(a.a1, a.a2)));
    fp2 c1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.a2)), fp2_mul(a.a0, a.a1));
    fp2 c2 = fp2_sub(fp2_sqr(a.a1), fp2_mul(a.a0, a.a2));
    fp2 t = fp2_add(fp2_mul(a.a0, c0), fp2_mul_xi(fp2_add(fp2_mul(a.a2, c1), fp2_mul(a.a1, c2))));
    fp2 ti = fp2_inv(fp2_reduce(t));
    return fp6_reduce(fp6{fp2_mul(fp2_reduce(c0), ti), fp2_mul(fp2_reduce(c1), ti), fp2_mul(fp2_reduce(c2), ti)});
}

BLS_HD fp12 fp12_one() { return fp12{fp6{fp2_one(), fp2_zero(), fp2_zero()}, fp6_zero()}; }
BLS_HD fp12 fp12_conj(const fp12& a) { return fp12{a.c0, fp6_neg(a.c1)}; }

BLS_HD bool fp12_is_one(const fp12& a) {
    return fp2_eq(a.c0.a0, fp2_one()) & fp2_is_zero(a.c0.a1) & fp2_is_zero(a.c0.a2) & fp2_is_zero(a.c1.a0) &
           fp2_is_zero(a.c1.a1) & fp2_is_zero(a.c1.a2);
}

// 3 fp6 multiplications (54 fp mul)
BLS_HDN fp12 fp12_mul_impl(const fp12& a, const fp12& b) {
    fp6 t0 = fp6_mul(a.c0, b.c0);
    fp6 t1 = fp6_mul(a.c1, b.c1);
    fp6 c1 = fp6_sub_nc(fp6_sub_nc(fp6_mul(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1)), t0), t1);
    fp6 c0 = fp6_add_nc(t0, fp6_mul_by_v(t1));
    return fp12{fp6_reduce(c0), fp6_reduce(c1)};
}

// complex squaring: 2 fp6 multiplications
BLS_HD fp12 fp12_mul(fp12 a, fp12 b) { return fp12_mul_impl(a, b); }

BLS_HDN fp12 fp12_sqr_impl(const fp12& a) {
    fp6 t = fp6_mul(a.c0, a.c1);
    fp6 s = fp6_mul(fp6_add(a.c0, a.c1), fp6_add(a.c0, fp6_mul_by_v(a.c1)));
    fp6 c0 = fp6_sub_nc(fp6_sub_nc(s, t), fp6_mul_by_v(t));
    return fp12{fp6_reduce(c0), fp6_reduce(fp6_add_nc(t, t))};
}

BLS_HD fp12 fp12_sqr(fp12 a) { return fp12_sqr_impl(a); }

// Miller-loop line  l = l0 + l1*v + l2*v*w  (coefficients at tower slots c0.a0, c0.a1, c1.a1)
struct line_t {
    fp2 l0, l1, l2;
};

// f * line: 13 fp2 multiplications (Karatsuba over w: t0 = f.c0 (l0 + l1 v), t1 = f.c1 (l2 v),
// s = (f.c0 + f.c1)(l0 + (l1 + l2) v); each sparse Fp6 product is 5 resp. 3 Fp2 products).
// The sums are limb-wise: multiplier operands may hold 2 limb units, everything else up to 7, so only the two
// three-term operand sums and the 12 result coefficients take a carry step (18 instead of 40 per line).
// Inputs: f and the line coefficients carried (at most 1 limb unit).  The result is carried, NOT reduced:
// every coefficient is a sum of at most 8 products (|v| < 16p), which is stable under repeated multiplication
// by reduced lines.  Reduce (fp12_reduce) before a general fp12_mul.
// The line operands are reached through a provider so that the device can keep them parked in LDS:
//   L.mul_l0(x) = x * l0, L.mul_l1, L.mul_l2, L.mul_m1 (m1 = l1 + l2), L.mul_l0l1 (x * (l0 + l1)), L.mul_l0m1.
template <class LineOps>
BLS_MID fp12 fp12_mul_by_line_ops(const fp12& f, const LineOps& L) {
    const fp6 &a = f.c0, &b = f.c1;
    fp2 a0l0 = L.mul_l0(a.a0), a1l1 = L.mul_l1(a.a1);
    fp2 t0c1 = fp2_sub_nc(fp2_sub_nc(L.mul_l0l1(fp2_add_nc(a.a0, a.a1)), a0l0), a1l1);                           // 3 units
    fp2 t0c0 = fp2_add_nc(a0l0, fp2_mul_xi_nc(L.mul_l1(a.a2)));                                                 // 3
    fp2 t0c2 = fp2_add_nc(a1l1, L.mul_l0(a.a2));                                                                // 2
    fp2 t1a0 = fp2_mul_xi(L.mul_l2(b.a2)), t1a1 = L.mul_l2(b.a0), t1a2 = L.mul_l2(b.a1);                         // 1 each
    fp2 s0 = fp2_add_nc(a.a0, b.a0), s1 = fp2_add_nc(a.a1, b.a1), s2 = fp2_add_nc(a.a2, b.a2);
    fp2 s0l0 = L.mul_l0(s0), s1m1 = L.mul_m1(s1);
    fp2 sc1 = fp2_sub_nc(fp2_sub_nc(L.mul_l0m1(fp2_carry(fp2_add_nc(s0, s1))), s0l0), s1m1);                     // 3
    fp2 sc0 = fp2_add_nc(s0l0, fp2_mul_xi_nc(L.mul_m1(s2)));                                                    // 3
    fp2 sc2 = fp2_add_nc(s1m1, L.mul_l0(s2));                                                                   // 2
    fp12 r;
    r.c1.a0 = fp2_carry(fp2_sub_nc(fp2_sub_nc(sc0, t0c0), t1a0));             // c1 = s - t0 - t1: 3 + 3 + 1
    r.c1.a1 = fp2_carry(fp2_sub_nc(fp2_sub_nc(sc1, t0c1), t1a1));
    r.c1.a2 = fp2_carry(fp2_sub_nc(fp2_sub_nc(sc2, t0c2), t1a2));
    r.c0.a0 = fp2_carry(fp2_add_nc(t0c0, fp2_mul_xi_nc(t1a2)));               // c0 = t0 + v t1
    r.c0.a1 = fp2_carry(fp2_add_nc(t0c1, t1a0));
    r.c0.a2 = fp2_carry(fp2_add_nc(t0c2, t1a1));
    return r;
}
// line operands held in registers
struct line_ops_regs {
    line_t l;
    fp2 m1;
    BLS_HD fp2 mul_l0(const fp2& x) const { return fp2_mul(x, l.l0); }
    BLS_HD fp2 mul_l1(const fp2& x) const { return fp2_mul(x, l.l1); }
    BLS_HD fp2 mul_l2(const fp2& x) const { return fp2_mul(x, l.l2); }
    BLS_HD fp2 mul_m1(const fp2& x) const { return fp2_mul(x, m1); }
    BLS_HD fp2 mul_l0l1(const fp2& x) const { return fp2_mul(x, fp2_add_nc(l.l0, l.l1)); }
    BLS_HD fp2 mul_l0m1(const fp2& x) const { return fp2_mul(x, fp2_carry(fp2_add_nc(l.l0, m1))); }
};
BLS_MID fp12 fp12_mul_by_line_karatsuba(const fp12& f, const line_t& l) {
    return fp12_mul_by_line_ops(f, line_ops_regs{l, fp2_add_nc(l.l1, l.l2)});
}
#if defined(__HIP_DEVICE_COMPILE__)
// line operands parked in four LDS slots of the kernel (l0, l1, l2, l1 + l2) by park(): they hold no registers
// between their uses; the two operand sums are rebuilt from LDS right before their single use
struct line_ops_lds {
    bls_lds_u32x4* base;           // 4 slots
    __device__ __forceinline__ void park(const line_t& l) const {
        fp2_lds_put(base, l.l0);
        fp2_lds_put(base + BLS_LDS_SLOT, l.l1);
        fp2_lds_put(base + 2 * BLS_LDS_SLOT, l.l2);
        fp2_lds_put(base + 3 * BLS_LDS_SLOT, fp2_add_nc(l.l1, l.l2));
    }
    __device__ __forceinline__ fp2 mul_l0(const fp2& x) const { return fp2_mul_lds(x, base); }
    __device__ __forceinline__ fp2 mul_l1(const fp2& x) const { return fp2_mul_lds(x, base + BLS_LDS_SLOT); }
    __device__ __forceinline__ fp2 mul_l2(const fp2& x) const { return fp2_mul_lds(x, base + 2 * BLS_LDS_SLOT); }
    __device__ __forceinline__ fp2 mul_m1(const fp2& x) const { return fp2_mul_lds(x, base + 3 * BLS_LDS_SLOT); }
    __device__ __forceinline__ fp2 mul_l0l1(const fp2& x) const { return fp2_mul(x, fp2_add_nc(fp2_lds_get(base), fp2_lds_get(base + BLS_LDS_SLOT))); }
    __device__ __forceinline__ fp2 mul_l0m1(const fp2& x) const {
        return fp2_mul(x, fp2_carry(fp2_add_nc(fp2_lds_get(base), fp2_lds_get(base + 3 * BLS_LDS_SLOT))));
    }
};
#else
// host pass of a .hip translation unit: kernels are parsed, never run
struct line_ops_lds : line_ops_regs {
    void park(const line_t&) const {}
};
#endif
// f * line as a SCHOOLBOOK product with one reduction per output coefficient (fp_dotn_core): with f = (a0 + a1 v + a2 v^2) +
// (b0 + b1 v + b2 v^2) w and the line l0 + l1 v + l2 v w (v^3 = xi, w^2 = v)
//   c0.a0 = a0 l0 + xi a2 l1 + xi b1 l2      c1.a0 = b0 l0 + xi b2 l1 + xi a2 l2
//   c0.a1 = a1 l0 +    a0 l1 + xi b2 l2      c1.a1 = b1 l0 +    b0 l1 +    a0 l2
//   c0.a2 = a2 l0 +    a1 l1 +    b0 l2      c1.a2 = b2 l0 +    b1 l1 +    a1 l2
// every coefficient an Fp2 sum of three products = two Fp dot products of six terms: 12 x (6 x 196 + 196) = 16 464 multiply-adds and 12
// reductions where the Karatsuba form (fp12_mul_by_line_ops) takes 15 288 and 26 reductions plus ~1 500 instructions of operand sums,
// differences and carry steps: ~17 600 instructions per line instead of ~21 000.  The xi multiples are limb-wise (real part a
// difference of canonical limbs: 1 unit; imaginary part a sum: 2 units), so the column bound of fp_dotn holds with exactly 8 units.
// Inputs: f with canonical limbs and |v| < 2p per coefficient (what this function returns; fp12_from_line of a stored line), line
// coefficients as k_lines stores them (|v| < 2p, limbs within one unit).  Output: canonical limbs, |v| < 2p: stable under iteration,
// and already below the bound fp12_mul wants, so no fp12_reduce is needed behind it.
BLS_HD fp2 fp2_dot3(const fp2& X, const fp2& Y, const fp2& Z, const fp2& L0, const fp2& L1, const fp2& L2) {
    const fp y[6] = {L0.c0, L0.c1, L1.c0, L1.c1, L2.c0, L2.c1};
    const fp xr[6] = {X.c0, fp_neg(X.c1), Y.c0, fp_neg(Y.c1), Z.c0, fp_neg(Z.c1)};
    const fp xi[6] = {X.c1, X.c0, Y.c1, Y.c0, Z.c1, Z.c0};
    return fp2{fp_dotn<6>(xr, y), fp_dotn<6>(xi, y)};
}
// xi * a for a with canonical limbs: (a0 - a1, a0 + a1), no carry step
BLS_HD fp2 fp2_mul_xi_pos(const fp2& a) { return fp2{fp_sub_pos(a.c0, a.c1), fp_add_nc(a.c0, a.c1)}; }
BLS_MID fp12 fp12_mul_by_line_lazy(const fp12& f, const line_t& l) {
    const fp6 &a = f.c0, &b = f.c1;
    const fp2 xa2 = fp2_mul_xi_pos(a.a2), xb1 = fp2_mul_xi_pos(b.a1), xb2 = fp2_mul_xi_pos(b.a2);
    fp12 r;
    r.c0.a0 = fp2_dot3(a.a0, xa2, xb1, l.l0, l.l1, l.l2);
    r.c0.a1 = fp2_dot3(a.a1, a.a0, xb2, l.l0, l.l1, l.l2);
    r.c0.a2 = fp2_dot3(a.a2, a.a1, b.a0, l.l0, l.l1, l.l2);
    r.c1.a0 = fp2_dot3(b.a0, xb2, xa2, l.l0, l.l1, l.l2);
    r.c1.a1 = fp2_dot3(b.a1, b.a0, a.a0, l.l0, l.l1, l.l2);
    r.c1.a2 = fp2_dot3(b.a2, b.a1, a.a1, l.l0, l.l1, l.l2);
    return r;
}
BLS_MID fp12 fp12_mul_by_line(const fp12& f, const line_t& l) { return fp12_mul_by_line_lazy(f, l); }
BLS_HD fp12 fp12_reduce(const fp12& a) { return fp12{fp6_reduce(a.c0), fp6_reduce(a.c1)}; }

BLS_HD fp12 fp12_from_line(const line_t& l) {
    return fp12{fp6{l.l0, l.l1, fp2_zero()}, fp6{fp2_zero(), l.l2, fp2_zero()}};
}

BLS_HDN fp12 fp12_inv(const fp12& a) {
    fp6 d = fp6_reduce(fp6_sub(fp6_mul(a.c0, a.c0), fp6_mul_by_v(fp6_mul(a.c1, a.c1))));
    fp6 di = fp6_inv(d);
    return fp12{fp6_reduce(fp6_mul(a.c0, di)), fp6_reduce(fp6_neg(fp6_mul(a.c1, di)))};
}

// a^p.  Flat basis w^i <-> tower: w^0,2,4 = c0.a0,a1,a2 ; w^1,3,5 = c1.a0,a1,a2
BLS_HDN fp12 fp12_frob(const fp12& a) {
    fp12 r;
    r.c0.a0 = fp2_conj(a.c0.a0);
    r.c1.a0 = fp2_mul(fp2_conj(a.c1.a0), fp2_from_const(k::FROB_G1));
    r.c0.a1 = fp2_mul(fp2_conj(a.c0.a1), fp2_from_const(k::FROB_G2));
    r.c1.a1 = fp2_mul(fp2_conj(a.c1.a1), fp2_from_const(k::FROB_G3));
    r.c0.a2 = fp2_mul(fp2_conj(a.c0.a2), fp2_from_const(k::FROB_G4));
    r.c1.a2 = fp2_mul(fp2_conj(a.c1.a2), fp2_from_const(k::FROB_G5));
    return r;
}

// a^(p^2)
BLS_HDN fp12 fp12_frob2(const fp12& a) {
    fp12 r;
    r.c0.a0 = a.c0.a0;
    r.c1.a0 = fp2_mul_fp(a.c1.a0, fp_from_const(k::FROB2_G1));
    r.c0.a1 = fp2_mul_fp(a.c0.a1, fp_from_const(k::FROB2_G2));
    r.c1.a1 = fp2_mul_fp(a.c1.a1, fp_from_const(k::FROB2_G3));
    r.c0.a2 = fp2_mul_fp(a.c0.a2, fp_from_const(k::FROB2_G4));
    r.c1.a2 = fp2_mul_fp(a.c1.a2, fp_from_const(k::FROB2_G5));
    return r;
}

BLS_HD void fp12_store_le(uint8_t* p, const fp12& a) {
    fp2_store_le(p, a.c0.a0);
    fp2_store_le(p + 96, a.c0.a1);
    fp2_store_le(p + 192, a.c0.a2);
    fp2_store_le(p + 288, a.c1.a0);
    fp2_store_le(p + 384, a.c1.a1);
    fp2_store_le(p + 480, a.c1.a2);
}

BLS_HD fp12 fp12_load_le(const uint8_t* p) {
    fp12 a;
    a.c0.a0 = fp2_load_le(p);
    a.c0.a1 = fp2_load_le(p + 96);
    a.c0.a2 = fp2_load_le(p + 192);
    a.c1.a0 = fp2_load_le(p + 288);
    a.c1.a1 = fp2_load_le(p + 384);
    a.c1.a2 = fp2_load_le(p + 480);
    return a;
}

}  // namespace bls
