// Short-Weierstrass arithmetic y^2 = x^3 + b (a = 0) in Jacobian coordinates, generic over the
// coordinate field: G1 over fp (blst_p1 / blst_p1_affine), G2 over fp2 (blst_p2 / blst_p2_affine).
// Infinity: Jacobian Z = 0; affine all-zero (the reference's vec_is_zero convention,
// blst_lowlevel.nim:28-44).
#pragma once
#include "fp.hpp"

namespace bls {

// field-generic spellings
BLS_HD fp f_add(const fp& a, const fp& b) { return fp_add(a, b); }
BLS_HD fp f_sub(const fp& a, const fp& b) { return fp_sub(a, b); }
BLS_HD fp f_mul(const fp& a, const fp& b) { return fp_mul(a, b); }
BLS_HD fp f_sqr(const fp& a) { return fp_sqr(a); }
BLS_HD fp f_neg(const fp& a) { return fp_neg(a); }
BLS_HD fp f_dbl(const fp& a) { return fp_dbl(a); }
BLS_HD bool f_is_zero(const fp& a) { return fp_is_zero(a); }
BLS_HD fp f_select(bool c, const fp& a, const fp& b) { return fp_select(c, a, b); }
BLS_HD fp2 f_add(const fp2& a, const fp2& b) { return fp2_add(a, b); }
BLS_HD fp2 f_sub(const fp2& a, const fp2& b) { return fp2_sub(a, b); }
BLS_HD fp2 f_mul(const fp2& a, const fp2& b) { return fp2_mul(a, b); }
BLS_HD fp2 f_sqr(const fp2& a) { return fp2_sqr(a); }
BLS_HD fp2 f_neg(const fp2& a) { return fp2_neg(a); }
BLS_HD fp2 f_dbl(const fp2& a) { return fp2_dbl(a); }
BLS_HD bool f_is_zero(const fp2& a) { return fp2_is_zero(a); }
BLS_HD fp2 f_select(bool c, const fp2& a, const fp2& b) { return fp2_select(c, a, b); }
// limb-wise (carry-less) variants: the caller carries (f_carry) or reduces (f_red) the end of the chain;
// at most 7 limb units may pile up, and anything fed to a multiplication must be carried (see fp.hpp)
BLS_HD fp f_add_nc(const fp& a, const fp& b) { return fp_add_nc(a, b); }
BLS_HD fp f_sub_nc(const fp& a, const fp& b) { return fp_sub_nc(a, b); }
BLS_HD fp f_dbl_nc(const fp& a) { return fp_dbl_nc(a); }
BLS_HD fp f_carry(const fp& a) { return fp_carry(a); }
BLS_HD fp2 f_add_nc(const fp2& a, const fp2& b) { return fp2_add_nc(a, b); }
BLS_HD fp2 f_sub_nc(const fp2& a, const fp2& b) { return fp2_sub_nc(a, b); }
BLS_HD fp2 f_dbl_nc(const fp2& a) { return fp2_dbl_nc(a); }
BLS_HD fp2 f_carry(const fp2& a) { return fp2_carry(a); }
// partial reduction (|v| < 0.51p): applied to stored coordinates so value bounds never accumulate
BLS_HD fp f_red(const fp& a) { return fp_reduce(a); }
BLS_HD fp2 f_red(const fp2& a) { return fp2_reduce(a); }
// a b - c d with ONE reduction where the field has a lazily reduced dot product (Fp: fp_dot2, 588 multiply-adds + 68 instead of
// 2 x (392 + 68) and a subtraction with its carry); Fp2: two products.  Operands: at most 2 limb units each.  Result carried.
BLS_HD fp f_mul_sub_mul(const fp& a, const fp& b, const fp& c, const fp& d) { return fp_dot2(a, b, fp_neg(c), d); }
BLS_HD fp2 f_mul_sub_mul(const fp2& a, const fp2& b, const fp2& c, const fp2& d) { return fp2_carry(fp2_sub_nc(fp2_mul(a, b), fp2_mul(c, d))); }
template <class F>
BLS_HD F f_zero();
template <>
BLS_HD fp f_zero<fp>() { return fp_zero(); }
template <>
BLS_HD fp2 f_zero<fp2>() { return fp2_zero(); }
template <class F>
BLS_HD F f_one();
template <>
BLS_HD fp f_one<fp>() { return fp_one(); }
template <>
BLS_HD fp2 f_one<fp2>() { return fp2_one(); }

template <class F>
struct aff {
    F x, y;
};
template <class F>
struct jac {
    F x, y, z;
};
using g1_aff = aff<fp>;
using g1_jac = jac<fp>;
using g2_aff = aff<fp2>;
using g2_jac = jac<fp2>;

template <class F>
BLS_HD bool aff_is_inf(const aff<F>& p) { return f_is_zero(p.x) & f_is_zero(p.y); }
template <class F>
BLS_HD bool jac_is_inf(const jac<F>& p) { return f_is_zero(p.z); }
template <class F>
BLS_HD jac<F> jac_inf() { return jac<F>{f_zero<F>(), f_zero<F>(), f_zero<F>()}; }
template <class F>
BLS_HD jac<F> jac_from_aff(const aff<F>& p) {
    bool inf = aff_is_inf(p);
    return jac<F>{p.x, p.y, f_select(inf, f_zero<F>(), f_one<F>())};
}
template <class F>
BLS_HD jac<F> jac_neg(const jac<F>& p) { return jac<F>{p.x, f_neg(p.y), p.z}; }
template <class F>
BLS_HD jac<F> jac_select(bool c, const jac<F>& a, const jac<F>& b) {
    return jac<F>{f_select(c, a.x, b.x), f_select(c, a.y, b.y), f_select(c, a.z, b.z)};
}

// Who multiplies: the shared out-of-line multiplier bodies (mul_shared: every formula's default), or the bodies expanded in place
// (mul_inplace: G2 only, see fp2_mul_inl)
struct mul_shared {
    template <class F> BLS_HD F mul(const F& a, const F& b) const { return f_mul(a, b); }
    template <class F> BLS_HD F sqr(const F& a) const { return f_sqr(a); }
};
struct mul_inplace {
    BLS_HD fp2 mul(const fp2& a, const fp2& b) const { return fp2_mul_inl(a, b); }
    BLS_HD fp2 sqr(const fp2& a) const { return fp2_sqr_inl(a); }
    BLS_HD fp mul(const fp& a, const fp& b) const { return fp_mul_inl(a, b); }
    BLS_HD fp sqr(const fp& a) const { return fp_sqr_inl(a); }
};
// dbl-2009-l (a = 0): 2M + 5S.  Z = 0 or Y = 0 give Z3 = 0.
template <class F, class M>
BLS_MID jac<F> jac_dbl_m(const jac<F>& p, const M& m) {
    F A = m.sqr(p.x);
    F B = m.sqr(p.y);
    F C = m.sqr(B);
    F D = f_carry(f_dbl_nc(f_sub_nc(f_sub_nc(m.sqr(f_add(p.x, B)), A), C)));     // 6 limb units, one carry
    F E = f_carry(f_add_nc(f_dbl_nc(A), A));
    F Fq = m.sqr(E);
    jac<F> r;
    r.x = f_red(f_sub_nc(Fq, f_dbl_nc(D)));
    F C8 = f_dbl_nc(f_carry(f_dbl_nc(f_dbl_nc(C))));
    r.y = f_carry(f_sub_nc(m.mul(E, f_sub_nc(D, r.x)), C8));     // y, z: carried only (|y| < 19p, |z| < 5p)
    r.z = f_carry(f_dbl_nc(m.mul(p.y, p.z)));
    return r;
}
template <class F>
BLS_MID jac<F> jac_dbl(const jac<F>& p) { return jac_dbl_m(p, mul_shared{}); }

// The same doubling for G2 with the multiplier bodies in place (the doubling runs of the cofactor clearing) and Y3 as ONE lazily
// reduced pair: Y3 = E (D - X3) - 8 B^2, so C = B^2 is never formed on its own (D = 4 X B directly):
//   A = X^2, B = Y^2, D = 4 X B, E = 3 A, X3 = E^2 - 2 D, Z3 = 2 Y Z,
//   Y3.re = E0 W0 - E1 W1 - [8 (B0 + B1)] (B0 - B1),   Y3.im = E0 W1 + E1 W0 - [8 B0] [2 B1]      (W = D - X3)
// 3 squarings + 2 products + 2 three-term dot products = 6 272 multiply-adds as before, 12 reductions instead of 14 and none of the
// (X + B)^2 - A - C and 8 C glue: ~370 instructions fewer per doubling.
BLS_MID jac<fp2> jac_dbl_lazy(const jac<fp2>& p) {
    const mul_inplace m{};
    fp2 A = m.sqr(p.x);
    fp2 B = m.sqr(p.y);
    fp2 D = fp2_carry(fp2_dbl_nc(fp2_dbl_nc(m.mul(p.x, B))));                      // 4 X B
    fp2 E = fp2_carry(fp2_add_nc(fp2_dbl_nc(A), A));
    fp2 Fq = m.sqr(E);
    jac<fp2> r;
    r.x = f_red(fp2_sub_nc(Fq, fp2_dbl_nc(D)));
    fp2 W = fp2_sub_nc(D, r.x);
    const fp bs4 = fp_carry(fp_dbl_nc(fp_add_nc(B.c0, B.c1)));                     // 2 (B0 + B1), carried (1 unit)
    const fp bs8 = fp_carry(fp_dbl_nc(fp_dbl_nc(bs4)));                            // 8 (B0 + B1), carried
    const fp bd = fp_sub_pos(B.c0, B.c1);                                          // B0 - B1 (canonical limbs: a fresh square)
    const fp b08 = fp_carry(fp_dbl_nc(fp_dbl_nc(fp_carry(fp_dbl_nc(B.c0)))));      // 8 B0, carried
    const fp b12 = fp_dbl_nc(B.c1);                                                // 2 B1 (2 units)
    const fp xr[3] = {E.c0, fp_neg(E.c1), fp_neg(bs8)}, yr[3] = {W.c0, W.c1, bd};
    const fp xi[3] = {E.c0, E.c1, fp_neg(b08)}, yi[3] = {W.c1, W.c0, b12};
    r.y = fp2{fp_dotn<3>(xr, yr), fp_dotn<3>(xi, yi)};
    r.z = fp2_carry(fp2_dbl_nc(m.mul(p.y, p.z)));
    return r;
}

// Lane-cooperative doubling for latency-bound chains (one point, or a few, and a whole wave to spend): a TEAM of lanes that all
// hold the same point computes the independent products of each round of dbl-2009-l in ONE multiplier call, every lane taking
// one of them, and shares the results.  The formula, its carries and its reductions are jac_dbl's; only who multiplies differs,
// and that is the Team's business: team_solo (below; host tests and the bounds tracker) computes every product itself, the
// device teams of kernels.hip select an operand pair by role and broadcast the results with wave shuffles.
struct team_solo {
    template <class F>
    BLS_HD void mul3(F& r0, F& r1, F& r2, const F& a0, const F& b0, const F& a1, const F& b1, const F& a2, const F& b2) const {
        r0 = f_mul(a0, b0); r1 = f_mul(a1, b1); r2 = f_mul(a2, b2);
    }
    template <class F>
    BLS_HD void sqr3(F& r0, F& r1, F& r2, const F& a0, const F& a1, const F& a2) const {
        r0 = f_sqr(a0); r1 = f_sqr(a1); r2 = f_sqr(a2);
    }
    template <class F>
    BLS_HD F mul1(const F& a, const F& b) const { return f_mul(a, b); }
    template <class F>
    BLS_HD void mul2(F& r0, F& r1, const F& a0, const F& b0, const F& a1, const F& b1) const {
        r0 = f_mul(a0, b0); r1 = f_mul(a1, b1);
    }
    template <class F>
    BLS_HD void sqr2(F& r0, F& r1, const F& a0, const F& a1) const {
        r0 = f_sqr(a0); r1 = f_sqr(a1);
    }
    template <class F>
    BLS_HD void mul4(F& r0, F& r1, F& r2, F& r3, const F& a0, const F& b0, const F& a1, const F& b1, const F& a2, const F& b2, const F& a3, const F& b3) const {
        r0 = f_mul(a0, b0); r1 = f_mul(a1, b1); r2 = f_mul(a2, b2); r3 = f_mul(a3, b3);
    }
};
template <class F, class Team>
BLS_MID jac<F> jac_dbl_team(const jac<F>& p, const Team& team) {
    F A, B, YZ;
    team.mul3(A, B, YZ, p.x, p.x, p.y, p.y, p.y, p.z);                               // X^2 | Y^2 | Y Z
    F E = f_carry(f_add_nc(f_dbl_nc(A), A));
    F C, t, Fq;
    team.sqr3(C, t, Fq, B, f_add(p.x, B), E);                                        // B^2 | (X+B)^2 | E^2
    F D = f_carry(f_dbl_nc(f_sub_nc(f_sub_nc(t, A), C)));
    jac<F> r;
    r.x = f_red(f_sub_nc(Fq, f_dbl_nc(D)));
    F C8 = f_dbl_nc(f_carry(f_dbl_nc(f_dbl_nc(C))));
    r.y = f_carry(f_sub_nc(team.mul1(E, f_sub_nc(D, r.x)), C8));
    r.z = f_carry(f_dbl_nc(YZ));
    return r;
}

// Jacobian + affine, complete (handles infinity operands, P == Q, P == -Q).
template <class F>
BLS_MID jac<F> jac_add_aff(const jac<F>& p, const aff<F>& q) {
    bool p_inf = jac_is_inf(p);
    bool q_inf = aff_is_inf(q);
    F Z1Z1 = f_sqr(p.z);
    F U2 = f_mul(q.x, Z1Z1);
    F S2 = f_mul(f_mul(q.y, p.z), Z1Z1);
    F H = f_sub(U2, p.x);
    F rr = f_sub(S2, p.y);
    bool h0 = f_is_zero(H), r0 = f_is_zero(rr);
    if (!p_inf && !q_inf && h0 && r0) return jac_dbl(p);
    F HH = f_sqr(H);
    F HHH = f_mul(H, HH);
    F V = f_mul(p.x, HH);
    jac<F> r;
    r.x = f_red(f_sub_nc(f_sub_nc(f_sqr(rr), HHH), f_dbl_nc(V)));
    r.y = f_mul_sub_mul(rr, f_sub_nc(V, r.x), p.y, HHH);
    r.z = f_mul(p.z, H);   // = 0 when P == -Q
    r = jac_select(q_inf, p, r);
    r = jac_select(p_inf, jac_from_aff(q), r);
    return r;
}

// Extended Jacobian coordinates ("XYZZ": x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2; infinity: ZZ = 0) for accumulators
// that only ever take mixed additions - the Pippenger buckets: madd-2008-s is 8M + 2S where the Jacobian mixed addition above
// is 8M + 3S (no Z1^2 to form).  Complete like jac_add_aff: infinity operands, P == Q (mdbl-2008-s-1 from the affine operand),
// P == -Q (PP = 0 gives ZZ3 = 0).  (X ZZ, Y ZZZ, ZZ) is the same point in Jacobian coordinates: two multiplications, once per
// bucket.
template <class F>
struct xyzz {
    F x, y, zz, zzz;
};
template <class F>
BLS_HD xyzz<F> xyzz_inf() { return xyzz<F>{f_zero<F>(), f_zero<F>(), f_zero<F>(), f_zero<F>()}; }
template <class F>
BLS_HD xyzz<F> xyzz_select(bool c, const xyzz<F>& a, const xyzz<F>& b) {
    return xyzz<F>{f_select(c, a.x, b.x), f_select(c, a.y, b.y), f_select(c, a.zz, b.zz), f_select(c, a.zzz, b.zzz)};
}
template <class F>
BLS_MID xyzz<F> xyzz_dbl_aff(const aff<F>& q) {          // q not at infinity
    F U = f_dbl(q.y);
    F V = f_sqr(U);
    F W = f_mul(U, V);
    F S = f_mul(q.x, V);
    F XX = f_sqr(q.x);
    F M = f_carry(f_add_nc(f_dbl_nc(XX), XX));
    xyzz<F> r;
    r.x = f_red(f_sub_nc(f_sqr(M), f_dbl_nc(S)));
    r.y = f_mul_sub_mul(M, f_sub_nc(S, r.x), W, q.y);
    r.zz = V;
    r.zzz = W;
    return r;
}
// q_inf: "q is the point at infinity", known to the caller (xyzz_add_aff tests the coordinates)
template <class F>
BLS_MID xyzz<F> xyzz_add_aff_flag(const xyzz<F>& p, const aff<F>& q, bool q_inf) {
    bool p_inf = f_is_zero(p.zz);
    F U2 = f_mul(q.x, p.zz);
    F S2 = f_mul(q.y, p.zzz);
    F P = f_sub(U2, p.x);
    F R = f_sub(S2, p.y);
    bool p0 = f_is_zero(P), r0 = f_is_zero(R);
    if (!p_inf && !q_inf && p0 && r0) return xyzz_dbl_aff(q);
    F PP = f_sqr(P);
    F PPP = f_mul(P, PP);
    F Q = f_mul(p.x, PP);
    xyzz<F> r;
    r.x = f_red(f_sub_nc(f_sub_nc(f_sqr(R), PPP), f_dbl_nc(Q)));
    r.y = f_mul_sub_mul(R, f_sub_nc(Q, r.x), p.y, PPP);
    r.zz = f_mul(p.zz, PP);       // = 0 when P == -Q
    r.zzz = f_mul(p.zzz, PPP);
    r = xyzz_select(q_inf, p, r);
    F one_or_zero = f_select(q_inf, f_zero<F>(), f_one<F>());
    r = xyzz_select(p_inf, xyzz<F>{q.x, q.y, one_or_zero, one_or_zero}, r);
    return r;
}
template <class F>
BLS_MID xyzz<F> xyzz_add_aff(const xyzz<F>& p, const aff<F>& q) { return xyzz_add_aff_flag(p, q, aff_is_inf(q)); }
template <class F>
BLS_MID jac<F> jac_from_xyzz(const xyzz<F>& p) { return jac<F>{f_mul(p.x, p.zz), f_mul(p.y, p.zzz), p.zz}; }

// Jacobian + Jacobian, complete.  Out of line (jac_add_impl takes references, i.e. memory operands); callers go
// through the by-value wrapper jac_add below so that only its private copies have their address taken and
// the caller's own (often loop-carried) points stay in registers.
template <class F>
BLS_MID jac<F> jac_add_body(const jac<F>& p, const jac<F>& q) {
    bool p_inf = jac_is_inf(p);
    bool q_inf = jac_is_inf(q);
    F Z1Z1 = f_sqr(p.z);
    F Z2Z2 = f_sqr(q.z);
    F U1 = f_mul(p.x, Z2Z2);
    F U2 = f_mul(q.x, Z1Z1);
    F S1 = f_mul(f_mul(p.y, q.z), Z2Z2);
    F S2 = f_mul(f_mul(q.y, p.z), Z1Z1);
    F H = f_sub(U2, U1);
    F rr = f_sub(S2, S1);
    bool h0 = f_is_zero(H), r0 = f_is_zero(rr);
    if (!p_inf && !q_inf && h0 && r0) return jac_dbl(p);
    F HH = f_sqr(H);
    F HHH = f_mul(H, HH);
    F V = f_mul(U1, HH);
    jac<F> r;
    r.x = f_red(f_sub_nc(f_sub_nc(f_sqr(rr), HHH), f_dbl_nc(V)));
    r.y = f_mul_sub_mul(rr, f_sub_nc(V, r.x), S1, HHH);
    r.z = f_mul(f_mul(p.z, q.z), H);
    r = jac_select(q_inf, p, r);
    r = jac_select(p_inf, q, r);
    return r;
}

template <class F>
BLS_HDN jac<F> jac_add_impl(const jac<F>& p, const jac<F>& q) { return jac_add_body(p, q); }
template <class F>
BLS_HD jac<F> jac_add(jac<F> p, jac<F> q) { return jac_add_impl(p, q); }

// acc + base for a base that is added again and again (the doubling chains of the cofactor clearing add the same point five times per
// chain): Z2^2 and Z2^3 are computed once (jac_precompute), so an addition is 3 squarings + 11 products instead of 4 + 12.  NOT complete:
// an operand at infinity or P == +-Q all give Z3 = 0 (Z3 = Z1 Z2 H) with X3, Y3 meaningless - the caller tests Z3 after every addition
// (and the base's Z once) and recomputes with the complete formulas if it ever is zero.  These are the formulas of the assembly loop of
// k_hash_clear (tools/gen_clear_asm.py); the host test build runs them under the bounds tracker.
template <class F>
struct jac_pre {
    jac<F> p;
    F zz, zzz;
};
template <class F>
BLS_MID jac_pre<F> jac_precompute(const jac<F>& q) {
    F zz = f_sqr(q.z);
    return jac_pre<F>{q, zz, f_mul(q.z, zz)};
}
template <class F>
BLS_MID jac<F> jac_add_pre(const jac<F>& p, const jac_pre<F>& q) {
    F Z1Z1 = f_sqr(p.z);
    F U2 = f_mul(q.p.x, Z1Z1);
    F S2 = f_mul(f_mul(q.p.y, p.z), Z1Z1);
    F Z12 = f_mul(p.z, q.p.z);
    F U1 = f_mul(p.x, q.zz);
    F H = f_sub(U2, U1);
    F S1 = f_mul(p.y, q.zzz);
    F rr = f_sub(S2, S1);
    jac<F> r;
    r.z = f_mul(Z12, H);
    F HH = f_sqr(H);
    F HHH = f_mul(H, HH);
    F V = f_mul(U1, HH);
    r.x = f_red(f_sub_nc(f_sub_nc(f_sqr(rr), HHH), f_dbl_nc(V)));
    r.y = f_carry(f_sub_nc(f_mul(rr, f_sub_nc(V, r.x)), f_mul(S1, HHH)));
    return r;
}

// Lane-cooperative complete addition (the formulas, carries and reductions of jac_add_body; products by the Team): five rounds
//   Z1^2, Z2^2  |  X1 Z2Z2, X2 Z1Z1, Y1 Z2, Y2 Z1  |  (Y1 Z2) Z2Z2, (Y2 Z1) Z1Z1, Z1 Z2, H^2  |  H HH, U1 HH, (Z1 Z2) H, r^2  |  r (V - X3), S1 HHH
// instead of sixteen multiplications in a row on every lane.  P == Q falls through to the team doubling.
template <class F, class Team>
BLS_MID jac<F> jac_add_team(const jac<F>& p, const jac<F>& q, const Team& team) {
    bool p_inf = jac_is_inf(p);
    bool q_inf = jac_is_inf(q);
    F Z1Z1, Z2Z2;
    team.sqr2(Z1Z1, Z2Z2, p.z, q.z);
    F U1, U2, T1, T2;
    team.mul4(U1, U2, T1, T2, p.x, Z2Z2, q.x, Z1Z1, p.y, q.z, q.y, p.z);
    F H = f_sub(U2, U1);
    F S1, S2, Z12, HH;
    team.mul4(S1, S2, Z12, HH, T1, Z2Z2, T2, Z1Z1, p.z, q.z, H, H);
    F rr = f_sub(S2, S1);
    bool h0 = f_is_zero(H), r0 = f_is_zero(rr);
    if (!p_inf && !q_inf && h0 && r0) return jac_dbl_team(p, team);
    F HHH, V, Z3, RR;
    team.mul4(HHH, V, Z3, RR, H, HH, U1, HH, Z12, H, rr, rr);
    jac<F> r;
    r.x = f_red(f_sub_nc(f_sub_nc(RR, HHH), f_dbl_nc(V)));
    F A, B;
    team.mul2(A, B, rr, f_sub_nc(V, r.x), S1, HHH);
    r.y = f_carry(f_sub_nc(A, B));
    r.z = Z3;
    r = jac_select(q_inf, p, r);
    r = jac_select(p_inf, q, r);
    return r;
}

// [k]P for a 64-bit scalar, affine base; left-to-right double-and-add.  Not constant time: the
// blinding scalars are public (blst_min_pubkey_sig_core.nim:531-541 rationale).
template <class F>
BLS_HDN jac<F> jac_mul_u64(const aff<F>& p, uint64_t kk) {
    jac<F> acc = jac_inf<F>();
    for (int i = 63; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((kk >> i) & 1) acc = jac_add_aff(acc, p);
    }
    return acc;
}

// [k]P for a 64-bit scalar with signed 4-bit windows: k = sum d_j 16^j, d_j in [-8, 8].  In a wave the lanes hold
// different scalars, so the bit-serial form above executes its conditional addition in (almost) every one of the
// 64 iterations - some lane always has the bit set; here every lane adds once per window: 64 doublings +
// 17 additions + a table of 1..8 times P (4 doublings, 3 mixed additions) instead of 64 + 64.
template <class F>
BLS_MID jac<F> jac_mul_u64_w4_body(const aff<F>& p, uint64_t kk) {
    // T[i] = (i + 1) P.  One copy of the doubling and of the mixed addition in a loop: unrolled, this function
    // was 84 KB of code against a 64 KB instruction cache.
    jac<F> T[8];
    T[0] = jac_from_aff(p);
#pragma clang loop unroll(disable)
    for (int i = 1; i < 8; i++) {
        if (i & 1)
            T[i] = jac_dbl(T[i >> 1]);              // 2, 4, 6, 8 times P
        else
            T[i] = jac_add_aff(T[i - 1], p);        // 3, 5, 7 times P
    }
    // signed digits from the least significant end; dig[16] is the final carry (0 or 1)
    int8_t dig[17];
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int32_t d = (int32_t)((kk >> (4 * j)) & 15u) + (int32_t)carry;
        carry = d > 8;
        dig[j] = (int8_t)(carry ? d - 16 : d);
    }
    dig[16] = (int8_t)carry;
    jac<F> acc = jac_select(carry != 0, T[0], jac_inf<F>());
#pragma clang loop unroll(disable)
    for (int j = 15; j >= 0; j--) {
#pragma clang loop unroll(disable)
        for (int k4 = 0; k4 < 4; k4++) acc = jac_dbl(acc);
        int d = dig[j];
        if (d != 0) {
            jac<F> t = T[(d < 0 ? -d : d) - 1];
            if (d < 0) t = jac_neg(t);
            acc = jac_add_body(acc, t);          // inlined: one copy in this loop, operands stay in registers
        }
    }
    return acc;
}
// out-of-line form (its own register budget); k_pkmul inlines the body so that the kernel's launch bounds govern it
template <class F>
BLS_HDN jac<F> jac_mul_u64_w4(const aff<F>& p, uint64_t kk) { return jac_mul_u64_w4_body(p, kk); }

// [k]P for Jacobian base
template <class F>
BLS_HDN jac<F> jac_mul_u64_jac(const jac<F>& p, uint64_t kk) {
    jac<F> acc = jac_inf<F>();
    for (int i = 63; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((kk >> i) & 1) acc = jac_add_body(acc, p);      // inlined (one copy): no operands through memory
    }
    return acc;
}

// [k]P for a 256-bit scalar (8 little-endian words).  NOT constant time: only the bench/test input generator
// (k_sign_*) uses it, never a production signer.
template <class F>
BLS_HDN jac<F> jac_mul_256(const aff<F>& p, const uint32_t (&kk)[8]) {
    jac<F> acc = jac_inf<F>();
    for (int i = 255; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((kk[i >> 5] >> (i & 31)) & 1) acc = jac_add_aff(acc, p);
    }
    return acc;
}
template <class F>
BLS_HDN jac<F> jac_mul_256_jac(const jac<F>& p, const uint32_t (&kk)[8]) {
    jac<F> acc = jac_inf<F>();
    for (int i = 255; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((kk[i >> 5] >> (i & 31)) & 1) acc = jac_add_body(acc, p);
    }
    return acc;
}

BLS_HD g1_aff g1_aff_load(const uint8_t* p) { return g1_aff{fp_load_le(p), fp_load_le(p + 48)}; }
BLS_HD g2_aff g2_aff_load(const uint8_t* p) { return g2_aff{fp2_load_le(p), fp2_load_le(p + 96)}; }
BLS_HD void g1_jac_store(uint8_t* p, const g1_jac& a) {
    fp_store_le(p, a.x);
    fp_store_le(p + 48, a.y);
    fp_store_le(p + 96, a.z);
}
BLS_HD g1_jac g1_jac_load(const uint8_t* p) { return g1_jac{fp_load_le(p), fp_load_le(p + 48), fp_load_le(p + 96)}; }
BLS_HD void g2_jac_store(uint8_t* p, const g2_jac& a) {
    fp2_store_le(p, a.x);
    fp2_store_le(p + 96, a.y);
    fp2_store_le(p + 192, a.z);
}
BLS_HD g2_jac g2_jac_load(const uint8_t* p) { return g2_jac{fp2_load_le(p), fp2_load_le(p + 96), fp2_load_le(p + 192)}; }

}  // namespace bls
