// Optimal-ate pairing pieces for BLS12-381 (x = -0xd201000000010000), organised "line-major":
//   1. per pair (P in G1, Q in G2): walk T = Q through the 63 doublings + 5 additions of |x| and
//      emit the 68 line functions evaluated at P (miller_lines);
//   2. per step s: L_s = prod over pairs of line_{pair,s}   (embarrassingly parallel);
//   3. f = Horner over s of (f^2 * L_s), conjugate, final exponentiation (one per batch).
// This replaces blst_miller_loop_n / blst_pairing_commit / blst_pairing_merge /
// blst_pairing_finalverify (reference blst_abi.nim:455-509; call sites
// blst_min_pubkey_sig_core.nim:649-672).  The shared Fp12 squarings of BLST's N_MAX=8 batches become
// ONE squaring chain per batch; Q and P stay projective, so no inversion is spent per tuple.
//
// Twist convention: E2: y^2 = x^3 + 4(1+u) (M-type), untwist (x,y) -> (x/w^2, y/w^3), w^6 = xi.
// A line through T with slope lam evaluated at P=(xp,yp), times w^3 (killed by the final exp):
//   (lam*xt - yt)  -  lam*xp * v  +  yp * v*w          -> tower slots c0.a0, c0.a1, c1.a1.
#pragma once
#include "curve.hpp"
#include "tower.hpp"

namespace bls {

constexpr int N_LINES = 68;

// P-side factors: for Jacobian P=(X,Y,Z), lines are scaled by Z^3: xp*Z^3 = X*Z, yp*Z^3 = Y.
struct g1_pre {
    fp z3, xz, y;
};

BLS_HD g1_pre g1_precompute(const g1_jac& p) {
    fp z2 = fp_sqr(p.z);
    return g1_pre{fp_mul(z2, p.z), fp_mul(p.x, p.z), p.y};
}

BLS_HD line_t line_scale(const fp2& c0, const fp2& c1, const fp2& c2, const g1_pre& p) {
    return line_t{fp2_mul_fp(c0, p.z3), fp2_mul_fp(c1, p.xz), fp2_mul_fp(c2, p.y)};
}

BLS_HD line_t line_one() { return line_t{fp2_one(), fp2_zero(), fp2_zero()}; }

// T <- 2T, returns tangent line at T evaluated at P.
BLS_MID line_t miller_dbl_step(g2_jac& t, const g1_pre& p) {
    fp2 A = fp2_sqr(t.x);
    fp2 B = fp2_sqr(t.y);
    fp2 C = fp2_sqr(B);
    fp2 D = fp2_carry(fp2_dbl_nc(fp2_sub_nc(fp2_sub_nc(fp2_sqr(fp2_add(t.x, B)), A), C)));
    fp2 E = fp2_carry(fp2_add_nc(fp2_dbl_nc(A), A));
    fp2 Fq = fp2_sqr(E);
    fp2 zz = fp2_sqr(t.z);
    fp2 x3 = fp2_reduce(fp2_sub_nc(Fq, fp2_dbl_nc(D)));
    fp2 y3 = fp2_sub_nc(fp2_mul(E, fp2_sub_nc(D, x3)), fp2_dbl_nc(fp2_carry(fp2_dbl_nc(fp2_dbl_nc(C)))));     // carried below
    fp2 z3 = fp2_dbl(fp2_mul(t.y, t.z));
    // line * (Z3 * Z^2):  (E*X - 2B)  -  E*Z^2 * xp v  +  Z3*Z^2 * yp vw
    fp2 c0 = fp2_carry(fp2_sub_nc(fp2_mul(E, t.x), fp2_dbl_nc(B)));
    fp2 c1 = fp2_neg(fp2_mul(E, zz));
    fp2 c2 = fp2_mul(z3, zz);
    t = g2_jac{x3, fp2_carry(y3), z3};
    return line_scale(c0, c1, c2, p);
}

// Q-side factors for the 5 addition steps (Q Jacobian)
struct g2_addpre {
    fp2 z2, z3;     // Zq^2, Zq^3
};

// T <- T + Q, returns chord line through T and Q evaluated at P.
BLS_MID line_t miller_add_step(g2_jac& t, const g2_jac& q, const g2_addpre& qp, const g1_pre& p) {
    fp2 Z1Z1 = fp2_sqr(t.z);
    fp2 U1 = fp2_mul(t.x, qp.z2);
    fp2 U2 = fp2_mul(q.x, Z1Z1);
    fp2 S1 = fp2_mul(t.y, qp.z3);
    fp2 S2 = fp2_mul(fp2_mul(q.y, t.z), Z1Z1);
    fp2 H = fp2_sub(U2, U1);
    fp2 rr = fp2_sub(S2, S1);
    fp2 HH = fp2_sqr(H);
    fp2 HHH = fp2_mul(H, HH);
    fp2 V = fp2_mul(U1, HH);
    fp2 x3 = fp2_reduce(fp2_sub_nc(fp2_sub_nc(fp2_sqr(rr), HHH), fp2_dbl_nc(V)));
    fp2 y3 = fp2_sub_nc(fp2_mul(rr, fp2_sub_nc(V, x3)), fp2_mul(S1, HHH));                                      // carried below
    fp2 z3 = fp2_mul(fp2_mul(t.z, q.z), H);
    // slope = rr / Z3.  line * (Z3 * Zq^3): (rr*Xq*Zq - Yq*Z3) - rr*Zq^3 * xp v + Z3*Zq^3 * yp vw
    fp2 c0 = fp2_sub(fp2_mul(rr, fp2_mul(q.x, q.z)), fp2_mul(q.y, z3));
    fp2 c1 = fp2_neg(fp2_mul(rr, qp.z3));
    fp2 c2 = fp2_mul(z3, qp.z3);
    t = g2_jac{x3, fp2_carry(y3), z3};
    return line_scale(c0, c1, c2, p);
}

// Emits the 68 lines of pair (P, Q) through sink(step, line).  A pair with P or Q at infinity
// contributes 1 (blst skips such pairs in the Miller loop).
template <class Sink>
BLS_HD void miller_lines(const g1_jac& pj, const g2_jac& q, Sink&& sink) {
    bool skip = jac_is_inf(pj) | jac_is_inf(q);
    g1_pre p = g1_precompute(pj);
    g2_addpre qp;
    qp.z2 = fp2_sqr(q.z);
    qp.z3 = fp2_mul(qp.z2, q.z);
    g2_jac t = q;
    int s = 0;
    for (int bit = 62; bit >= 0; bit--) {
        line_t l = miller_dbl_step(t, p);
        sink(s++, skip ? line_one() : l);
        if ((k::X_ABS >> bit) & 1) {
            line_t a = miller_add_step(t, q, qp, p);
            sink(s++, skip ? line_one() : a);
        }
    }
}

// f = conj( Horner_s (f^2 [at doubling steps] * L_s) ), L given in miller_lines step order.
template <class Src>
BLS_HD fp12 miller_combine(Src&& src) {
    fp12 f = fp12_one();
    int s = 0;
    for (int bit = 62; bit >= 0; bit--) {
        f = fp12_sqr(f);
        f = fp12_mul(f, src(s++));
        if ((k::X_ABS >> bit) & 1) f = fp12_mul(f, src(s++));
    }
    return fp12_conj(f);
}

// a^|x| then conjugate (x < 0), for a in the cyclotomic subgroup
BLS_HDN fp12 fp12_cyc_exp_x(const fp12& a) {
    fp12 r = a;
    for (int bit = 62; bit >= 0; bit--) {
        r = fp12_sqr(r);
        if ((k::X_ABS >> bit) & 1) r = fp12_mul(r, a);
    }
    return fp12_conj(r);
}

// f^(3 (p^12-1)/r): easy part, then 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3
BLS_HDN fp12 final_exp(const fp12& f) {
    fp12 t = fp12_mul(fp12_conj(f), fp12_inv(f));
    t = fp12_mul(fp12_frob2(t), t);
    fp12 a = fp12_mul(fp12_cyc_exp_x(t), fp12_conj(t));
    a = fp12_mul(fp12_cyc_exp_x(a), fp12_conj(a));
    fp12 b = fp12_mul(fp12_cyc_exp_x(a), fp12_frob(a));
    fp12 c = fp12_mul(fp12_mul(fp12_cyc_exp_x(fp12_cyc_exp_x(b)), fp12_frob2(b)), fp12_conj(b));
    return fp12_mul(c, fp12_mul(fp12_sqr(t), t));
}

}  // namespace bls
