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
//   (lam*xt - yt)  -  lam*xp * v  +  yp * v*w          -> tower slots c0.a0, c0.a1, c1.a1
// (any non-zero factor from Fp2 may be applied to a line: the final exponentiation removes it).
#pragma once
#include "curve.hpp"
#include "tower.hpp"

namespace bls {

constexpr int N_LINES = 68;

// P-side factors: for Jacobian P=(X,Y,Z), xp = X/Z^2 and yp = Y/Z^3; every line is scaled by Z^3 (an Fp
// factor, killed by the final exponentiation): constant term * Z^3, xp term * X Z, yp term * Y.
struct g1_pre {
    fp z3, xz, nxz3, y;      // Z^3, X Z, -3 X Z, Y
};

BLS_HD g1_pre g1_precompute(const g1_jac& p) {
    fp z2 = fp_sqr(p.z);
    fp xz = fp_mul(p.x, p.z);
    return g1_pre{fp_mul(z2, p.z), xz, fp_neg(fp_carry(fp_add_nc(fp_dbl_nc(xz), xz))), p.y};
}

BLS_HD line_t line_one() { return line_t{fp2_one(), fp2_zero(), fp2_zero()}; }

// The point T walking through the multiples of Q is kept in HOMOGENEOUS projective coordinates (x = X/Z,
// y = Y/Z): for y^2 = x^3 + b' the doubling together with its tangent line takes 2 multiplications and 7
// squarings in Fp2 (Costello, Lange, Naehrig, "Faster pairing computations on curves with high-degree twists",
// PKC 2010, a = 0 case) against 5 + 6 in Jacobian coordinates.  Lines are defined up to factors in Fp2 (a proper
// subfield of Fp12: such factors vanish in the final exponentiation), which is what lets the formulas drop
// every denominator.
using g2_proj = g2_jac;      // same three Fp2 coordinates, homogeneous meaning

// Jacobian (X, Y, Z) -> homogeneous (X Z : Y : Z^3)
BLS_HD g2_proj g2_to_proj(const g2_jac& q) {
    fp2 z2 = fp2_sqr(q.z);
    return g2_proj{fp2_mul(q.x, q.z), q.y, fp2_mul(z2, q.z)};
}

// T <- 2T, returns the tangent line at T evaluated at P, times 2 Y Z^2 (an Fp2 factor):
//   (Y^2 - 3b' Z^2)  -  3 X^2 * xp v  +  2 Y Z * yp vw          with b' = 4 xi
//   X3 = 2 X Y (B - 3E), Y3 = (B + 3E)^2 - 12 E^2, Z3 = 4 B H;  B = Y^2, C = Z^2, E = 3 b' C, H = 2 Y Z
// a^2 - 12 e^2 in Fp2 as TWO lazily reduced dot products (fp_dot2: one reduction each) instead of two squarings, a quadrupling, a
// tripling, a subtraction and a partial reduction:  re = (a0 + a1)(a0 - a1) - [4 (e0 + e1)] [3 (e0 - e1)],  im = [2 a0] a1 - [8 e0] [3 e1].
// a: carried (|v| < 4p per coefficient), e: reduced with canonical limbs (fp2_reduce output).  The scaled operands take one carry step
// each (the multiplier takes operands of at most 2 limb units).  Result: canonical limbs, |v| < 2p.
BLS_MID fp2 fp2_sqr_minus_12sqr(const fp2& a, const fp2& e) {
    const fp e4s = fp_dbl_nc(fp_carry(fp_dbl_nc(fp_add_nc(e.c0, e.c1))));              // 4 (e0 + e1): 4 limb units, carry, 2 units
    const fp ed = fp_sub_pos(e.c0, e.c1);
    const fp e3d = fp_carry(fp_add_nc(fp_dbl_nc(ed), ed));                              // 3 (e0 - e1)
    const fp e8 = fp_dbl_nc(fp_carry(fp_dbl_nc(fp_dbl_nc(e.c0))));                      // 8 e0: 4 units, carry, 2 units
    const fp e3 = fp_carry(fp_add_nc(fp_dbl_nc(e.c1), e.c1));                           // 3 e1
    return fp2{fp_dot2(fp_add_nc(a.c0, a.c1), fp_sub_nc(a.c0, a.c1), fp_neg(e4s), e3d), fp_dot2(fp_dbl_nc(a.c0), a.c1, fp_neg(e8), e3)};
}
// (Variants that were measured and dropped - the first squaring of a step expanded in place, the multiplier bodies in place, y3 as
// two separate squarings, the five addition steps out of line, the flat 63-iteration loop: profiles/r04_ab/ab_lines_*.txt.)
template <class M>
BLS_MID line_t miller_dbl_step_m(g2_proj& t, const g1_pre& p, const M& m) {
    fp2 B = m.sqr(t.y);
    fp2 C = m.sqr(t.z);
    fp2 X2 = m.sqr(t.x);
    fp2 C4 = fp2_dbl_nc(fp2_carry(fp2_dbl_nc(fp2_mul_xi_nc(C))));                       // 4 xi C   (2 -> 4 units, carry, 2)
    fp2 E = fp2_reduce(fp2_add_nc(fp2_dbl_nc(C4), C4));                                 // 12 xi C = 3 b' C   (6 units)
    fp2 F = fp2_add_nc(fp2_dbl_nc(E), E);                                                // 3E, 3 units
    fp2 H = fp2_carry(fp2_sub_nc(fp2_sub_nc(m.sqr(fp2_add(t.y, t.z)), B), C));           // 2 Y Z
    fp2 XY2 = fp2_carry(fp2_sub_nc(fp2_sub_nc(m.sqr(fp2_add(t.x, t.y)), X2), B));       // 2 X Y = (X + Y)^2 - X^2 - Y^2: a squaring for a product
    fp2 x3 = m.mul(XY2, fp2_carry(fp2_sub_nc(B, F)));
    fp2 y3 = fp2_sqr_minus_12sqr(fp2_carry(fp2_add_nc(B, F)), E);                       // (B + 3E)^2 - 12 E^2 with two reductions instead of four
    fp2 z3 = fp2_carry(fp2_dbl_nc(fp2_dbl_nc(m.mul(B, H))));
    t = g2_proj{x3, y3, z3};
    fp2 BE = fp2_sub_nc(B, E);
    return line_t{fp2{m.mul(BE.c0, p.z3), m.mul(BE.c1, p.z3)}, fp2{m.mul(X2.c0, p.nxz3), m.mul(X2.c1, p.nxz3)}, fp2{m.mul(H.c0, p.y), m.mul(H.c1, p.y)}};
}
BLS_MID line_t miller_dbl_step(g2_proj& t, const g1_pre& p) { return miller_dbl_step_m(t, p, mul_shared{}); }

// T <- T + Q (both homogeneous), returns the chord through T and Q evaluated at P, times (x2 - x1) Z1 Z2^2:
//   (u X2 - v Y2)  -  u Z2 * xp v  +  v Z2 * yp vw,      u = Y2 Z1 - Y1 Z2,  v = X2 Z1 - X1 Z2
BLS_MID line_t miller_add_step(g2_proj& t, const g2_proj& q, const g1_pre& p) {
    fp2 Y1Z2 = fp2_mul(t.y, q.z), X1Z2 = fp2_mul(t.x, q.z), Z1Z2 = fp2_mul(t.z, q.z);
    fp2 u = fp2_sub(fp2_mul(q.y, t.z), Y1Z2);
    fp2 v = fp2_sub(fp2_mul(q.x, t.z), X1Z2);
    fp2 uu = fp2_sqr(u), vv = fp2_sqr(v);
    fp2 vvv = fp2_mul(v, vv);
    fp2 R = fp2_mul(vv, X1Z2);
    fp2 A = fp2_carry(fp2_sub_nc(fp2_sub_nc(fp2_mul(uu, Z1Z2), vvv), fp2_dbl_nc(R)));
    fp2 x3 = fp2_mul(v, A);
    fp2 y3 = fp2_reduce(fp2_sub_nc(fp2_mul(u, fp2_sub_nc(R, A)), fp2_mul(vvv, Y1Z2)));
    fp2 z3 = fp2_mul(vvv, Z1Z2);
    fp2 c0 = fp2_carry(fp2_sub_nc(fp2_mul(u, q.x), fp2_mul(v, q.y)));
    fp2 c1 = fp2_mul(u, q.z);
    fp2 c2 = fp2_mul(v, q.z);
    t = g2_proj{x3, y3, z3};
    return line_t{fp2_mul_fp(c0, p.z3), fp2_neg(fp2_mul_fp(c1, p.xz)), fp2_mul_fp(c2, p.y)};
}

// Emits the 68 lines of pair (P, Q) through sink(step, line).  A pair with P or Q at infinity
// contributes 1 (blst skips such pairs in the Miller loop).
template <class Sink>
BLS_HD void miller_lines(const g1_jac& pj, const g2_jac& qj, Sink&& sink) {
    bool skip = jac_is_inf(pj) | jac_is_inf(qj);
    g1_pre p = g1_precompute(pj);
    g2_proj q = g2_to_proj(qj);
    q = g2_proj{fp2_reduce(q.x), fp2_reduce(q.y), fp2_reduce(q.z)};
    g2_proj t = q;
    int s = 0;
    // |x| has six set bits: the 63 doubling steps are six RUNS (1, 2, 3, 9, 32, 16) with an addition step behind each but the last.
    // The runs are an inner loop of their own so that the code of the five addition steps (as large as the doubling step's) lies
    // OUTSIDE the loop that executes 63 times: that loop and the shared multiplier bodies it calls are then ~43 KB instead of ~60 KB,
    // which the 64 KB instruction cache serves at a higher rate (tools/ubench_icache.hip; same steps, same lines, same order).
    uint64_t rest = k::X_ABS & ~(1ull << 63);
    int pos = 63;
#pragma clang loop unroll(disable)
    while (pos > 0) {
        const int next = rest ? 63 - __builtin_clzll(rest) : 0;
        int n = pos - next;
#pragma clang loop unroll(disable)
        do {
            line_t l = miller_dbl_step(t, p);
            sink(s++, skip ? line_one() : l);
        } while (--n > 0);
        if (rest) {
            line_t a = miller_add_step(t, q, p);
            sink(s++, skip ? line_one() : a);
            rest &= ~(1ull << next);
        }
        pos = next;
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
