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
BLS_HDN fp6 fp6_mul(const fp6& a, const fp6& b) {
    fp2 t0 = fp2_mul(a.a0, b.a0);
    fp2 t1 = fp2_mul(a.a1, b.a1);
    fp2 t2 = fp2_mul(a.a2, b.a2);
    fp2 c0 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.a1, a.a2), fp2_add(b.a1, b.a2)), t1), t2);
    c0 = fp2_add(t0, fp2_mul_xi(c0));
    fp2 c1 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.a0, a.a1), fp2_add(b.a0, b.a1)), t0), t1);
    c1 = fp2_add(c1, fp2_mul_xi(t2));
    fp2 c2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(a.a0, a.a2), fp2_add(b.a0, b.a2)), t0), t2);
    c2 = fp2_add(c2, t1);
    return fp6{c0, c1, c2};
}

// a * (l0 + l1 v): 5 fp2 multiplications
BLS_MID fp6 fp6_mul_by_01(const fp6& a, const fp2& l0, const fp2& l1) {
    fp2 t0 = fp2_mul(a.a0, l0);
    fp2 t1 = fp2_mul(a.a1, l1);
    // results keep at most 2 limb units (c1 is carried): the caller sums them limb-wise and reduces once
    fp2 c1 = fp2_carry(fp2_sub_nc(fp2_sub_nc(fp2_mul(fp2_add(a.a0, a.a1), fp2_add(l0, l1)), t0), t1));
    fp2 c0 = fp2_add_nc(t0, fp2_mul_xi(fp2_mul(a.a2, l1)));
    fp2 c2 = fp2_add_nc(t1, fp2_mul(a.a2, l0));
    return fp6{c0, c1, c2};
}

// a * (l1 v): 3 fp2 multiplications
BLS_HD fp6 fp6_mul_by_1(const fp6& a, const fp2& l1) {
    return fp6{fp2_mul_xi(fp2_mul(a.a2, l1)), fp2_mul(a.a0, l1), fp2_mul(a.a1, l1)};
}

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
    return fp2_eq_any(a.c0.a0, fp2_one()) & fp2_is_zero_any(a.c0.a1) & fp2_is_zero_any(a.c0.a2) & fp2_is_zero_any(a.c1.a0) &
           fp2_is_zero_any(a.c1.a1) & fp2_is_zero_any(a.c1.a2);
}

// 3 fp6 multiplications (54 fp mul)
BLS_HDN fp12 fp12_mul(const fp12& a, const fp12& b) {
    fp6 t0 = fp6_mul(a.c0, b.c0);
    fp6 t1 = fp6_mul(a.c1, b.c1);
    fp6 c1 = fp6_sub(fp6_sub(fp6_mul(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1)), t0), t1);
    fp6 c0 = fp6_add(t0, fp6_mul_by_v(t1));
    return fp12{fp6_reduce(c0), fp6_reduce(c1)};
}

// complex squaring: 2 fp6 multiplications
BLS_HDN fp12 fp12_sqr(const fp12& a) {
    fp6 t = fp6_mul(a.c0, a.c1);
    fp6 s = fp6_mul(fp6_add(a.c0, a.c1), fp6_add(a.c0, fp6_mul_by_v(a.c1)));
    fp6 c0 = fp6_sub(fp6_sub(s, t), fp6_mul_by_v(t));
    return fp12{fp6_reduce(c0), fp6_reduce(fp6_dbl(t))};
}

// Miller-loop line  l = l0 + l1*v + l2*v*w  (coefficients at tower slots c0.a0, c0.a1, c1.a1)
struct line_t {
    fp2 l0, l1, l2;
};

// f * line: 13 fp2 multiplications
BLS_MID fp12 fp12_mul_by_line(const fp12& f, const line_t& l) {
    fp6 t0 = fp6_mul_by_01(f.c0, l.l0, l.l1);
    fp6 t1 = fp6_mul_by_1(f.c1, l.l2);
    fp6 s = fp6_mul_by_01(fp6_add(f.c0, f.c1), l.l0, fp2_add(l.l1, l.l2));
    fp6 c1 = fp6_sub_nc(fp6_sub_nc(s, t0), t1);            // 2 + 2 + 1 limb units
    fp6 c0 = fp6_add_nc(t0, fp6_mul_by_v(t1));             // 2 + 1
    // Only carried, not reduced: every coefficient is a sum of at most 8 products (|v| < 16p), which is stable
    // under repeated multiplication by reduced lines.  Reduce (fp12_reduce) before a general fp12_mul.
    return fp12{fp6_carry(c0), fp6_carry(c1)};
}
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
