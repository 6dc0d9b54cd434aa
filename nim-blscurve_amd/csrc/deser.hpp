// Batched deserialisation + validation of ZCash-format compressed points on the device: replaces, per tuple,
// PublicKey.fromBytes / Signature.fromBytes (reference blscurve/blst/bls_sig_io.nim:42-58, 81-99):
//   blst_p1_uncompress + "not infinity" + blst_p1_affine_in_g1      (public keys, 48 bytes)
//   blst_p2_uncompress + blst_p2_affine_in_g2 (infinity allowed)     (signatures, 96 bytes)
// Encoding (tests/priv_to_pub.sage:59-65, tests/serialization.nim:19-45): big-endian x (for G2: x.c1 then
// x.c0), bit 383 = compressed, bit 382 = infinity, bit 381 = "y is the lexicographically larger root".
// Subgroup membership uses the endomorphism tests (M. Scott, "A note on group membership tests for G1, G2 and
// GT on BLS pairing-friendly curves", 2021), as BLST does:  G1: phi(P) == [-x^2]P,  G2: psi(P) == [x]P.
#pragma once
#include "curve.hpp"
#include "h2c.hpp"

namespace bls {

enum deser_status : uint8_t {
    DESER_OK = 0,
    DESER_PK_BAD_ENCODING = 1,   // flags / x >= p / x not on the curve
    DESER_PK_NOT_IN_G1 = 2,
    DESER_PK_INFINITY = 3,       // "Infinity public keys are not allowed" (bls_sig_io.nim:95-97)
    DESER_SIG_BAD_ENCODING = 4,
    DESER_SIG_NOT_IN_G2 = 5,
};

// 48 big-endian bytes -> integer limbs (flag bits cleared); returns false if the integer is >= p
BLS_HD bool fp_from_be48(fp& out, const uint8_t* b, bool clear_flags) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint8_t* q = b + 44 - 4 * i;
        w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    if (clear_flags) w[11] &= 0x1fffffffu;
    fp v = fp_relimb_from32(w);
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        uint32_t d = v.l[i] - k::P[i] - borrow;
        borrow = d >> 31;
    }
    out = fp_to_mont(v);
    return borrow != 0;            // v < p
}

// y > (p-1)/2 for the canonical integer y, i.e. 2y >= p
BLS_HD bool fp_is_lex_largest(const fp& y_mont) {
    fp y = fp_from_mont(y_mont);
    uint32_t borrow = 0, c = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        uint32_t t = 2 * y.l[i] + c;
        c = i < FP_N - 1 ? t >> 28 : 0;
        uint32_t lim = i < FP_N - 1 ? (t & FP_MASK) : t;
        uint32_t d = lim - k::P[i] - borrow;
        borrow = d >> 31;
    }
    return borrow == 0;
}

// status 0 ok (inf set when the encoding is the point at infinity), else bad encoding
BLS_HDN bool g1_uncompress(g1_aff& out, bool& inf, const uint8_t* b) {
    inf = false;
    out = g1_aff{fp_zero(), fp_zero()};
    uint8_t f = b[0];
    if (!(f & 0x80)) return false;                     // blst_p1_uncompress: the compressed bit must be set
    if (f & 0x40) {
        uint32_t any = f & 0x3f;
        for (int i = 1; i < 48; i++) any |= b[i];
        inf = true;
        return any == 0;
    }
    fp x;
    if (!fp_from_be48(x, b, true)) return false;
    fp rhs = fp_add(fp_mul(fp_sqr(x), x), fp_from_const(k::B1));
    rhs = fp_reduce(rhs);
    fp t = fp_recip_sqrt_pow(rhs);
    fp y = fp_mul(rhs, t);
    if (!fp_eq(fp_sqr(y), rhs)) return false;
    bool want_large = (f & 0x20) != 0;
    if (fp_is_lex_largest(y) != want_large) y = fp_reduce(fp_neg(y));
    out = g1_aff{x, y};
    return true;
}

BLS_HDN bool g2_uncompress(g2_aff& out, bool& inf, const uint8_t* b) {
    inf = false;
    out = g2_aff{fp2_zero(), fp2_zero()};
    uint8_t f = b[0];
    if (!(f & 0x80)) return false;
    if (f & 0x40) {
        uint32_t any = f & 0x3f;
        for (int i = 1; i < 96; i++) any |= b[i];
        inf = true;
        return any == 0;
    }
    fp2 x;
    if (!fp_from_be48(x.c1, b, true)) return false;
    if (!fp_from_be48(x.c0, b + 48, false)) return false;
    fp2 b2{fp_from_const(k::B1), fp_from_const(k::B1)};                  // 4(1 + u)
    fp2 rhs = fp2_reduce(fp2_add(fp2_mul(fp2_sqr(x), x), b2));
    fp2 y;
    if (!sqrt_ratio_fp2(y, rhs, fp2_one())) return false;                // not a square: x is not on the curve
    y = fp2_reduce(y);
    bool large = fp_is_zero(y.c1) ? fp_is_lex_largest(y.c0) : fp_is_lex_largest(y.c1);
    bool want_large = (f & 0x20) != 0;
    if (large != want_large) y = fp2_reduce(fp2_neg(y));
    out = g2_aff{x, y};
    return true;
}

// P in G1  <=>  phi(P) == [-x^2]P, phi(x, y) = (beta x, y).  P affine, on the curve, not infinity.
BLS_HDN bool g1_in_subgroup(const g1_aff& p) {
    g1_jac t = jac_mul_u64(p, k::X_ABS);                                 // [|x|]P
    t = jac_mul_u64_jac(t, k::X_ABS);                                    // [x^2]P
    if (jac_is_inf(t)) return false;
    // compare with (beta x, -y): X == beta x Z^2, Y == -y Z^3
    fp z2 = fp_sqr(t.z), z3 = fp_mul(z2, t.z);
    fp ex = fp_mul(fp_mul(fp_from_const(k::BETA), p.x), z2);
    fp ey = fp_neg(fp_mul(p.y, z3));
    return fp_eq(t.x, ex) & fp_eq(t.y, ey);
}

// Q in G2  <=>  psi(Q) == [x]Q.  Q affine, on the curve, not infinity.
BLS_HDN bool g2_in_subgroup(const g2_aff& q) {
    g2_jac t = jac_neg(jac_mul_u64(q, k::X_ABS));                        // [x]Q, x < 0
    if (jac_is_inf(t)) return false;
    g2_jac ps = g2_psi(jac_from_aff(q));                                 // Z = 1
    fp2 z2 = fp2_sqr(t.z), z3 = fp2_mul(z2, t.z);
    return fp2_eq(t.x, fp2_mul(ps.x, z2)) & fp2_eq(t.y, fp2_mul(ps.y, z3));
}

// blst_p1_deserialize / blst_p2_deserialize (blst_abi.nim:394,400; used by fromBytes for 96- / 192-byte input,
// bls_sig_io.nim:49-52,88-91) [blst-upstream semantics]: the three top bits of byte 0 select the form:
//   000  uncompressed: big-endian x then y (G2: x.c1, x.c0, y.c1, y.c0), each < p, and the point must be on the curve;
//   1xx  compressed: the first half is a compressed encoding (the rest is not read);
//   01x  infinity: only valid as 0x40 followed by zeros;   anything else is a bad encoding.
BLS_HDN bool g1_deserialize(g1_aff& out, bool& inf, const uint8_t* b) {
    uint8_t f = b[0];
    if (f & 0x80) return g1_uncompress(out, inf, b);
    inf = false;
    out = g1_aff{fp_zero(), fp_zero()};
    if (f & 0x40) {
        uint32_t any = f & 0x3f;
        for (int i = 1; i < 96; i++) any |= b[i];
        inf = true;
        return any == 0;
    }
    if (f & 0x20) return false;
    fp x, y;
    if (!fp_from_be48(x, b, false) || !fp_from_be48(y, b + 48, false)) return false;
    fp rhs = fp_add(fp_mul(fp_sqr(x), x), fp_from_const(k::B1));
    if (!fp_eq(fp_sqr(y), rhs)) return false;                            // BLST_POINT_NOT_ON_CURVE
    out = g1_aff{x, y};
    return true;
}
BLS_HDN bool g2_deserialize(g2_aff& out, bool& inf, const uint8_t* b) {
    uint8_t f = b[0];
    if (f & 0x80) return g2_uncompress(out, inf, b);
    inf = false;
    out = g2_aff{fp2_zero(), fp2_zero()};
    if (f & 0x40) {
        uint32_t any = f & 0x3f;
        for (int i = 1; i < 192; i++) any |= b[i];
        inf = true;
        return any == 0;
    }
    if (f & 0x20) return false;
    fp2 x, y;
    if (!fp_from_be48(x.c1, b, false) || !fp_from_be48(x.c0, b + 48, false)) return false;
    if (!fp_from_be48(y.c1, b + 96, false) || !fp_from_be48(y.c0, b + 144, false)) return false;
    fp2 b2{fp_from_const(k::B1), fp_from_const(k::B1)};                  // 4(1 + u)
    fp2 rhs = fp2_add(fp2_mul(fp2_sqr(x), x), b2);
    if (!fp2_eq(fp2_sqr(y), rhs)) return false;
    out = g2_aff{x, y};
    return true;
}

// flags of the general form (include/blscurve_mi355x.h): which wire form each side has, and fromBytesKnownOnCurve
constexpr uint32_t DESER_F_PK_UNCOMPRESSED = 1, DESER_F_SIG_UNCOMPRESSED = 2, DESER_F_KNOWN_ON_CURVE = 4;

// One tuple: wire-format (pk, sig) -> validated affine points.  Returns the deser_status.
//   fromBytes (bls_sig_io.nim:42-58, 81-99): decode, "public key is not infinity", subgroup checks (infinity signature allowed)
//   fromBytesKnownOnCurve (:60-79, 101-121): the same without the subgroup checks
BLS_HD uint8_t deserialize_tuple(g1_aff& pk, g2_aff& sig, const uint8_t* pkb, const uint8_t* sigb, uint32_t flags) {
    bool inf;
    bool ok = (flags & DESER_F_PK_UNCOMPRESSED) ? g1_deserialize(pk, inf, pkb) : g1_uncompress(pk, inf, pkb);
    if (!ok) return DESER_PK_BAD_ENCODING;
    if (inf) return DESER_PK_INFINITY;
    if (!(flags & DESER_F_KNOWN_ON_CURVE) && !g1_in_subgroup(pk)) return DESER_PK_NOT_IN_G1;
    ok = (flags & DESER_F_SIG_UNCOMPRESSED) ? g2_deserialize(sig, inf, sigb) : g2_uncompress(sig, inf, sigb);
    if (!ok) return DESER_SIG_BAD_ENCODING;
    if (!(flags & DESER_F_KNOWN_ON_CURVE) && !inf && !g2_in_subgroup(sig)) return DESER_SIG_NOT_IN_G2;
    return DESER_OK;
}

}  // namespace bls
