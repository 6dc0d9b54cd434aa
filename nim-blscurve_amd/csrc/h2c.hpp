// hash_to_curve for G2, suite BLS12381G2_XMD:SHA-256_SSWU_RO_ (RFC 9380), inversion-free:
// replaces blst_hash_to_g2 / the hash part of blst_pairing_chk_n_mul_n_aggr_pk_in_g1
// (reference blst_abi.nim:383, 504-507; call site blst_min_pubkey_sig_core.nim:558-568).
// Output stays Jacobian: the Miller loop consumes projective Q, so no field inversion per tuple.
#pragma once
#include "curve.hpp"
#include "sha256.hpp"

namespace bls {

// 64 big-endian bytes (as 16 BE words) -> Fp (Montgomery), value mod p.  hi, lo < 2^256 < p.
BLS_HD fp fp_from_be_words16(const uint32_t* wbe) {
    uint32_t hw[12], lw[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        hw[i] = i < 8 ? wbe[7 - i] : 0u;
        lw[i] = i < 8 ? wbe[15 - i] : 0u;
    }
    fp hi = fp_relimb_from32(hw), lo = fp_relimb_from32(lw);
    fp him = fp_mul(fp_to_mont(hi), fp_from_const(k::TWO256));
    return fp_add(him, fp_to_mont(lo));
}

// expand_message_xmd(msg, dst, 256) -> two Fp2 elements (hash_to_field, count = 2, m = 2, L = 64)
BLS_HDN void hash_to_field_fp2x2(fp2& u0, fp2& u1, const uint8_t* msg, uint32_t msg_len, const uint8_t* dst, uint32_t dst_len) {
    sha256_ctx c;
    uint32_t b0[8], bi[8];
    sha256_begin(c);
    sha256_zero_block(c);                               // Z_pad
    sha256_update(c, msg, msg_len);
    sha256_put(c, 0x01);                                // l_i_b_str = 256
    sha256_put(c, 0x00);
    sha256_put(c, 0x00);
    sha256_update(c, dst, dst_len);
    sha256_put(c, (uint8_t)dst_len);
    sha256_end(c, b0);
    uint32_t uni[64];
    for (int i = 1; i <= 8; i++) {
        sha256_begin(c);
        if (i == 1) {
            sha256_update_words(c, b0);
        } else {
            uint32_t x[8];
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = b0[j] ^ bi[j];
            sha256_update_words(c, x);
        }
        sha256_put(c, (uint8_t)i);
        sha256_update(c, dst, dst_len);
        sha256_put(c, (uint8_t)dst_len);
        sha256_end(c, bi);
        for (int j = 0; j < 8; j++) uni[(i - 1) * 8 + j] = bi[j];
    }
    u0.c0 = fp_from_be_words16(uni);
    u0.c1 = fp_from_be_words16(uni + 16);
    u1.c0 = fp_from_be_words16(uni + 32);
    u1.c1 = fp_from_be_words16(uni + 48);
}

// expand_message_xmd for 32-byte messages (every SignatureSet message, bls_batch_verifier.nim:42) with everything
// that does not depend on the message folded into constants per DST, built once on the host:
//   b_0 = H(Z_pad(64) | msg(32) | 0x0100 | 0x00 | DST | len):  state after Z_pad, block 2 = msg | tail[0..32), block 3
//   b_i = H((b_0 ^ b_(i-1))(32) | i | DST | len):              block 1 = x | tail_i[0..32), block 2 (the same for all i)
// 18 compressions per message and no byte-wise buffering.  Valid for 28 <= dst_len <= 83 (the tails then span exactly
// the block boundaries used here); other lengths take hash_to_field_fp2x2.
struct xmd32_consts {
    uint32_t h_zpad[8], b0_w8[8], b0_blk3[16], bi_w8[8][8], bi_blk2[16];
    uint32_t valid;
};
inline void xmd32_pack(uint32_t* w, const uint8_t* bytes, int nwords) {
    for (int i = 0; i < nwords; i++)
        w[i] = ((uint32_t)bytes[4 * i] << 24) | ((uint32_t)bytes[4 * i + 1] << 16) | ((uint32_t)bytes[4 * i + 2] << 8) | bytes[4 * i + 3];
}
inline xmd32_consts xmd32_precompute(const uint8_t* dst, uint32_t dst_len) {
    xmd32_consts c{};
    c.valid = dst_len >= 28 && dst_len <= 83;
    if (!c.valid) return c;
    uint8_t blk[192];
    uint32_t w[16];
    sha256_init(c.h_zpad);
    for (int i = 0; i < 16; i++) w[i] = 0;
    sha256_compress_core(c.h_zpad, w);                                   // Z_pad: 64 zero bytes
    // b_0: tail = 0x01 0x00 | 0x00 | DST | len, message length 64 + 32 + dst_len + 4 bytes
    for (int i = 0; i < 192; i++) blk[i] = 0;
    uint32_t t0 = dst_len + 4, tot0 = 96 + t0;
    blk[0] = 1;
    for (uint32_t i = 0; i < dst_len; i++) blk[3 + i] = dst[i];
    blk[3 + dst_len] = (uint8_t)dst_len;
    blk[t0] = 0x80;
    xmd32_pack(c.b0_w8, blk, 8);
    xmd32_pack(c.b0_blk3, blk + 32, 16);
    c.b0_blk3[15] = tot0 * 8;
    // b_i: tail_i = i | DST | len, message length 32 + dst_len + 2 bytes
    uint32_t t1 = dst_len + 2, tot1 = 32 + t1;
    for (int k = 0; k < 8; k++) {
        for (int i = 0; i < 192; i++) blk[i] = 0;
        blk[0] = (uint8_t)(k + 1);
        for (uint32_t i = 0; i < dst_len; i++) blk[1 + i] = dst[i];
        blk[1 + dst_len] = (uint8_t)dst_len;
        blk[t1] = 0x80;
        xmd32_pack(c.bi_w8[k], blk, 8);
        if (k == 0) {
            xmd32_pack(c.bi_blk2, blk + 32, 16);
            c.bi_blk2[15] = tot1 * 8;
        }
    }
    return c;
}
BLS_HD void hash_to_field_fp2x2_msg32(fp2& u0, fp2& u1, const uint32_t (&msg_be)[8], const xmd32_consts& c) {
    uint32_t b0[8], bi[8], w[16], uni[64];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        b0[i] = c.h_zpad[i];
        w[i] = msg_be[i];
        w[8 + i] = c.b0_w8[i];
    }
    sha256_compress(b0, w);
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = c.b0_blk3[i];
    sha256_compress(b0, w);
#pragma unroll
    for (int i = 0; i < 8; i++) bi[i] = 0;
    for (int k = 0; k < 8; k++) {
        uint32_t h[8];
        sha256_init(h);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w[i] = b0[i] ^ bi[i];                                        // i = 1: b_0 alone (bi starts as zero)
            w[8 + i] = c.bi_w8[k][i];
        }
        sha256_compress(h, w);
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = c.bi_blk2[i];
        sha256_compress(h, w);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bi[i] = h[i];
            uni[k * 8 + i] = h[i];
        }
    }
    u0.c0 = fp_from_be_words16(uni);
    u0.c1 = fp_from_be_words16(uni + 16);
    u1.c0 = fp_from_be_words16(uni + 32);
    u1.c1 = fp_from_be_words16(uni + 48);
}

// (is_square(N/D), y) with y = sqrt(N/D) if square, else sqrt(Z * N/D); Z = -(2+u), norm(Z) = 5.
// Two Fp exponentiations; the first also yields 1/norm(D).
// (`pw`: the exponentiation a^((p-3)/4) - fp_recip_sqrt_pow in every lane, or k_hash_one's row form, which computes the wave's two chains along DPP rows)
struct pow_in_lane {
    BLS_HD fp operator()(const fp& a) const { return fp_recip_sqrt_pow(a); }
};
template <class Pow>
BLS_MID bool sqrt_ratio_fp2_with(fp2& y, const fp2& N, const fp2& D, const Pow& pw) {
    fp nN = fp2_norm(N), nD = fp2_norm(D);
    fp M = fp_mul(nN, nD);
    fp t = pw(M);
    fp s = fp_mul(M, t);                         // s^2 = M (QR) or -M
    bool is_sq = fp_eq(fp_sqr(s), M);
    fp t2 = fp_sqr(t);                           // M * t^2 = +-1
    fp invM = fp_select(is_sq, t2, fp_neg(t2));
    fp invnD = fp_mul(nN, invM);
    fp2 g = fp2_mul_fp(fp2_mul(N, fp2_conj(D)), invnD);
    fp n = fp_mul(s, invnD);                     // sqrt(norm(g))
    fp2 gz = fp2_mul(g, fp2_from_const(k::SSWU_Z));
    fp nz = fp_mul(n, fp_from_const(k::SQRT_M5));
    g = fp2_select(is_sq, g, gz);
    n = fp_select(is_sq, n, nz);
    fp half = fp_from_const(k::HALF);
    fp d = fp_mul(fp_add(g.c0, n), half);
    d = fp_select(fp_is_zero(d), fp_reduce(g.c0), d);
    fp t3 = pw(d);
    fp x0 = fp_mul(d, t3);
    bool qr = fp_eq(fp_sqr(x0), d);
    fp bh = fp_mul(fp_mul(g.c1, half), t3);
    y.c0 = fp_select(qr, x0, bh);
    y.c1 = fp_select(qr, bh, fp_neg(x0));
    return is_sq;
}
BLS_MID bool sqrt_ratio_fp2(fp2& y, const fp2& N, const fp2& D) { return sqrt_ratio_fp2_with(y, N, D, pow_in_lane{}); }

// Simplified SWU onto E2': y^2 = x^3 + 240u x + 1012(1+u)  (RFC 9380 appendix F.2), Jacobian output.
template <class Pow>
BLS_MID g2_jac sswu_g2_with(const fp2& u, const Pow& pw) {
    const fp2 A = fp2_from_const(k::SSWU_A), B = fp2_from_const(k::SSWU_B), Z = fp2_from_const(k::SSWU_Z);
    fp2 tv1 = fp2_mul(Z, fp2_sqr(u));
    fp2 tv2 = fp2_add(fp2_sqr(tv1), tv1);
    fp2 xn = fp2_mul(B, fp2_add(tv2, fp2_one()));                 // x1 numerator
    fp2 xd = fp2_select(fp2_is_zero(tv2), fp2_from_const(k::SSWU_ZA), fp2_mul(A, fp2_neg(tv2)));
    fp2 xd2 = fp2_sqr(xd);
    fp2 D = fp2_mul(xd2, xd);
    fp2 N = fp2_add(fp2_mul(fp2_add(fp2_sqr(xn), fp2_mul(A, xd2)), xn), fp2_mul(B, D));
    fp2 y1;
    bool is_sq = sqrt_ratio_fp2_with(y1, N, D, pw);
    fp2 x2n = fp2_mul(tv1, xn);
    fp2 y2 = fp2_mul(fp2_mul(tv1, u), y1);
    fp2 x = fp2_select(is_sq, xn, x2n);
    fp2 y = fp2_select(is_sq, y1, y2);
    bool same = fp2_sgn0(u) == fp2_sgn0(y);
    y = fp2_select(same, y, fp2_neg(y));
    return g2_jac{fp2_mul(x, xd), fp2_mul(y, D), xd};
}
BLS_MID g2_jac sswu_g2(const fp2& u) { return sswu_g2_with(u, pow_in_lane{}); }

// 3-isogeny E2' -> E2 on Jacobian coordinates: (XN(X,Z^2), Y*YN(X,Z^2), Z*(X - xK Z^2))
BLS_MID g2_jac iso3_g2(const g2_jac& p) {
    fp2 z2 = fp2_sqr(p.z), z4 = fp2_sqr(z2), z6 = fp2_mul(z4, z2);
    fp2 xn = fp2_add(fp2_mul(fp2_from_const(k::ISO_XN3), p.x), fp2_mul(fp2_from_const(k::ISO_XN2), z2));
    xn = fp2_add(fp2_mul(xn, p.x), fp2_mul(fp2_from_const(k::ISO_XN1), z4));
    xn = fp2_add(fp2_mul(xn, p.x), fp2_mul(fp2_from_const(k::ISO_XN0), z6));
    fp2 yn = fp2_add(fp2_mul(fp2_from_const(k::ISO_YN3), p.x), fp2_mul(fp2_from_const(k::ISO_YN2), z2));
    yn = fp2_add(fp2_mul(yn, p.x), fp2_mul(fp2_from_const(k::ISO_YN1), z4));
    yn = fp2_add(fp2_mul(yn, p.x), fp2_mul(fp2_from_const(k::ISO_YN0), z6));
    fp2 d = fp2_sub(p.x, fp2_mul(fp2_from_const(k::ISO_XK), z2));
    return g2_jac{xn, fp2_mul(p.y, yn), fp2_mul(p.z, d)};
}

BLS_HD g2_jac g2_psi(const g2_jac& p) {
    return g2_jac{fp2_mul(fp2_conj(p.x), fp2_from_const(k::PSI_CX)), fp2_mul(fp2_conj(p.y), fp2_from_const(k::PSI_CY)), fp2_conj(p.z)};
}

// [x]P, x = -0xd201000000010000
BLS_HD g2_jac g2_mul_x(const g2_jac& p) { return jac_neg(jac_mul_u64_jac(p, k::X_ABS)); }

// Where the base point of a doubling chain waits between its five additions: a plain copy here; k_hash_clear parks it in LDS
// (three Fp2 slots), so that it holds no registers during the 63 doublings.
struct g2_park_regs {
    g2_jac v;
    BLS_HD void put(const g2_jac& a) { v = a; }
    BLS_HD g2_jac get() const { return v; }
};

// h_eff clearing, RFC 9380 appendix G.3 (Budroni-Pintore): [x^2 - x - 1]P + [x - 1]psi(P) + psi^2(2P)
//   = psi^2(2P) - psi(P) - [x]P - P  +  [x]([x]P + psi(P))
// written as ONE copy of the 63-doubling chain executed twice (pass 0: t1 = [x]P; pass 1: [x](t1 + psi(P))).
// |x| = 0xd201000000010000 has six set bits, so the chain is six RUNS of consecutive doublings (1, 2, 3, 9, 32, 16) with an addition
// of the base point between them.  A run is ONE call dbl_run(acc, n): the accumulator of a run is a loop-carried value inside that
// function and stays in registers for the whole run.  (Round 3 called a doubling functor once per doubling; hipcc kept that functor
// out of line - two call sites, 8 400 instructions - so the accumulator went through scratch memory on the way in and on the way
// out of EVERY doubling: four dependent memory passes of 84 words per doubling at one wave per SIMD, i.e. with nothing to hide them:
// 10 % of k_hash_clear's wave cycles were s_waitcnt.  Now it crosses memory 8 times per pass instead of 64.)
// dbl_run: n >= 1 doublings (jac_dbl, or the lane-cooperative jac_dbl_team of a kernel that has a team of lanes per point)
// add_in: the additions inside the chain (accumulator + parked base); add: the others.  The one-lane-per-point kernels pass the
// inlined body / the out-of-line addition, the kernels with a team of lanes per point the lane-cooperative jac_add_team.
// dbl_run has exactly ONE call site here (the loop over the runs), so that a caller may force it inline: inside a kernel the
// accumulator is then a plain SSA value.  (Out of line, its loop-carried point lives in the function's return slot, i.e. in scratch
// memory, and is stored and re-loaded around every doubling all the same.)  dbl1: one doubling, for 2P (not on the hot path).
template <class Pt, class Park, class DblRun, class Dbl1, class AddIn, class Add>
BLS_MID Pt clear_cofactor_g2_with(const Pt& p, Park& park, DblRun&& dbl_run, Dbl1&& dbl1, AddIn&& add_in, Add&& add) {
    Pt base = p, u = p, res = p;
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < 2; pass++) {
        park.put(base);
        Pt acc = base;                                       // bit 63 of |x|
        uint64_t rest = k::X_ABS & ~(1ull << 63);            // set bits still to come
        int pos = 63;
#pragma clang loop unroll(disable)
        while (pos > 0) {
            const int next = rest ? 63 - __builtin_clzll(rest) : 0;
            acc = dbl_run(acc, pos - next);                  // the one call site
            if (rest) {
                acc = add_in(acc, park.get());
                rest &= ~(1ull << next);
            }
            pos = next;
        }
        acc = jac_neg(acc);                                  // x < 0
        if (pass == 0) {
            Pt t2 = g2_psi(p);
            u = add(g2_psi(g2_psi(dbl1(p))), jac_neg(t2));               // psi^2(2P) - psi(P)
            u = add(u, jac_neg(acc));                                    // - [x]P
            u = add(u, jac_neg(p));                                      // - P
            base = add(acc, t2);                                         // [x]P + psi(P)
        } else {
            res = add(u, acc);
        }
    }
    return res;
}
// n >= 1 doublings with the point as a loop-carried value; ONE copy of the doubling in the loop (do-while: no peeled first iteration)
template <class F, class M>
BLS_MID jac<F> jac_dbl_n(const jac<F>& a, int n, const M& m) {
    jac<F> r = a;
#pragma clang loop unroll(disable)
    do {
        r = jac_dbl_m(r, m);
    } while (--n > 0);
    return r;
}
// G2 with the bodies in place: the doubling with the lazily reduced Y3 (curve.hpp jac_dbl_lazy)
BLS_MID jac<fp2> jac_dbl_n(const jac<fp2>& a, int n, const mul_inplace&) {
    jac<fp2> r = a;
#pragma clang loop unroll(disable)
    do {
        r = jac_dbl_lazy(r);
    } while (--n > 0);
    return r;
}
// Round 5: the chain acc = [|x|] base as ONE unit (k_hash_clear runs it as a hand-allocated assembly loop, tools/gen_clear_asm.py) and the
// clearing around it: `chain(base)` returns [|x|] base, everything else (psi maps, the six additions outside the chains, the one plain
// doubling) stays with the caller's complete formulas.
template <class Pt, class Chain, class Dbl1, class Add>
BLS_MID Pt clear_cofactor_g2_chain(const Pt& p, Chain&& chain, Dbl1&& dbl1, Add&& add) {
    Pt base = p, u = p, res = p;
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < 2; pass++) {
        Pt acc = jac_neg(chain(base));                                   // [x] base (x < 0); the one call site of the chain
        if (pass == 0) {
            Pt t2 = g2_psi(p);
            u = add(g2_psi(g2_psi(dbl1(p))), jac_neg(t2));               // psi^2(2P) - psi(P)
            u = add(u, jac_neg(acc));                                    // - [x]P
            u = add(u, jac_neg(p));                                      // - P
            base = add(acc, t2);                                         // [x]P + psi(P)
        } else {
            res = add(u, acc);
        }
    }
    return res;
}
// What the chain computes, in the formulas of the assembly loop: the base normalised (partial reductions), its Z^2 and Z^3 once, 63 doublings
// (jac_dbl_lazy) and 5 incomplete additions (jac_add_pre); ok = false when a Z turned out zero on the way (an operand at infinity, P == +-Q):
// the result is then meaningless and the caller recomputes with complete formulas.  Host test build: bounds tracker + multiply-add census.
BLS_MID g2_jac g2_chain_x_fast(const g2_jac& b0, bool& ok) {
    const g2_jac base{fp2_reduce(b0.x), fp2_reduce(b0.y), fp2_reduce(b0.z)};
    ok = !fp2_is_zero(base.z);
    const jac_pre<fp2> pre = jac_precompute(base);
    g2_jac acc = base;
#pragma clang loop unroll(disable)
    for (int i = 62; i >= 0; i--) {
        acc = jac_dbl_lazy(acc);
        if ((k::X_ABS >> i) & 1) {
            acc = jac_add_pre(acc, pre);
            ok = ok & !fp2_is_zero(acc.z);
        }
    }
    return acc;
}
BLS_HD g2_jac g2_chain_x(const g2_jac& base) {
    bool ok;
    g2_jac r = g2_chain_x_fast(base, ok);
    return ok ? r : jac_mul_u64_jac(base, k::X_ABS);
}
#if defined(__HIP_DEVICE_COMPILE__)
#define BLS_LAMBDA_INLINE __attribute__((always_inline))
#else
#define BLS_LAMBDA_INLINE
#endif
template <class Park, class M>
BLS_MID g2_jac clear_cofactor_g2_with(const g2_jac& p, Park& park, const M& m) {
    return clear_cofactor_g2_with(
        p, park, [&](const g2_jac& a, int n) BLS_LAMBDA_INLINE { return jac_dbl_n(a, n, m); }, [](const g2_jac& a) { return jac_dbl(a); },
        [](const g2_jac& a, const g2_jac& b) { return jac_add_body(a, b); }, [](const g2_jac& a, const g2_jac& b) { return jac_add(a, b); });
}
BLS_HDN g2_jac clear_cofactor_g2(const g2_jac& p) {
    g2_park_regs park;
#if defined(__HIP_DEVICE_COMPILE__)
    return clear_cofactor_g2_with(p, park, mul_shared{});      // the out-of-line form (k_hash_var, the signer): compact code on the shared multipliers
#else
    (void)park;
    return clear_cofactor_g2_chain(p, [](const g2_jac& b) { return g2_chain_x(b); }, [](const g2_jac& a) { return jac_dbl(a); },
                                   [](const g2_jac& a, const g2_jac& b) { return jac_add(a, b); });   // host (tests/host_emu, bounds tracker, census): the formulas k_hash_clear runs
#endif
}

BLS_HDN g2_jac hash_to_g2(const uint8_t* msg, uint32_t msg_len, const uint8_t* dst, uint32_t dst_len) {
    fp2 u0, u1;
    hash_to_field_fp2x2(u0, u1, msg, msg_len, dst, dst_len);
    // one copy of the (inlined) map code, executed for u0 then u1
    fp2 u[2] = {u0, u1};
    g2_jac q[2];
#pragma clang loop unroll(disable)
    for (int j = 0; j < 2; j++) q[j] = iso3_g2(sswu_g2(u[j]));
    return clear_cofactor_g2(jac_add(q[0], q[1]));
}

}  // namespace bls
