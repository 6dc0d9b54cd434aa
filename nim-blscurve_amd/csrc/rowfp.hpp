// Limb-parallel Fp arithmetic for the DEPENDENT G1 chains at the end of a Pippenger run (round 6): one Fp value = one 32-bit register, limb l of the value in
// lane l of a DPP row (16 lanes; lanes 14 and 15 hold zero), a wave = four rows.  Where one lane multiplies two 14-limb values with 392 dependent multiply-adds
// (~0.9 us), a row does it in 14 steps of eleven instructions (~0.4 us): step k adds a_k * b (a_k broadcast over the row: row_newbcast, b rotated: row_ror - so
// that lane l always meets the term of column l, or of column l + 16 once column l has been consumed), makes column k divisible by 2^28 with the Montgomery
// digit m_k = -col_k / p mod 2^28 (m_k * p rotated the same way) and passes every lane's bits above 2^28 to its right-hand neighbour, cyclically: the consumed
// lane k restarts as column k + 16 with the carry of column k + 15.  The four rows of a wave take the independent products of a point formula (a G1 doubling is
// three such rounds, a Jacobian addition five), ds_bpermute hands the results round.  Values are the product's own: signed 28-bit limbs, Montgomery R = 2^392.
//
// The same source runs on the CPU (BLS_ROW_EMU: `rw` is then all 64 lanes of a wave, every row operation a loop) under tests/host_emu with the bounds asserted.
// Reference: blst_p1s_mult_pippenger's result (blst_abi.nim:336-340); the formulas are dbl-2009-l and add-2007-bl's unscaled form, as curve.hpp's team versions.
#pragma once
#include "curve.hpp"

namespace bls {

#if defined(BLS_ROW_EMU)
struct rw { int32_t l[64]; };
struct rw64 { int64_t l[64]; };
#define ROW_FN static inline
#define ROW_EACH for (int i = 0; i < 64; i++)
ROW_FN rw operator+(const rw& a, const rw& b) { rw r; ROW_EACH { int64_t v = (int64_t)a.l[i] + b.l[i]; BLS_REQUIRE(v == (int32_t)v, "row add overflow"); r.l[i] = (int32_t)v; } return r; }
ROW_FN rw operator-(const rw& a, const rw& b) { rw r; ROW_EACH { int64_t v = (int64_t)a.l[i] - b.l[i]; BLS_REQUIRE(v == (int32_t)v, "row sub overflow"); r.l[i] = (int32_t)v; } return r; }
ROW_FN rw operator&(const rw& a, const rw& b) { rw r; ROW_EACH r.l[i] = a.l[i] & b.l[i]; return r; }
ROW_FN rw row_and(const rw& a, uint32_t m) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)a.l[i] & m); return r; }
ROW_FN rw row_shl(const rw& a, int n) { rw r; ROW_EACH { int64_t v = (int64_t)a.l[i] * ((int64_t)1 << n); BLS_REQUIRE(v == (int32_t)v, "row shl overflow"); r.l[i] = (int32_t)v; } return r; }
ROW_FN rw row_sar(const rw& a, int n) { rw r; ROW_EACH r.l[i] = a.l[i] >> n; return r; }
ROW_FN rw row_sext2(const rw& a) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)a.l[i] << 30) >> 30; return r; }
ROW_FN rw row_mullo(const rw& a, uint32_t c) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)a.l[i] * c); return r; }
ROW_FN rw64 row_mad(const rw& a, const rw& b, const rw64& acc) {
    rw64 r;
    ROW_EACH {
        __int128 v = (__int128)acc.l[i] + (__int128)a.l[i] * b.l[i];
        BLS_REQUIRE(v < ((__int128)1 << 62) && v > -((__int128)1 << 62), "row mad: accumulator beyond 2^62");
        r.l[i] = (int64_t)v;
    }
    return r;
}
ROW_FN rw64 row_zero64() { rw64 r; ROW_EACH r.l[i] = 0; return r; }
ROW_FN rw row_zero() { rw r; ROW_EACH r.l[i] = 0; return r; }
ROW_FN rw row_lo(const rw64& a) { rw r; ROW_EACH r.l[i] = (int32_t)(uint32_t)a.l[i]; return r; }
ROW_FN rw row_hi28(const rw64& a) { rw r; ROW_EACH { int64_t v = a.l[i] >> 28; BLS_REQUIRE(v == (int32_t)v, "row carry beyond 32 bits"); r.l[i] = (int32_t)v; } return r; }
ROW_FN rw64 row_ext(const rw& a) { rw64 r; ROW_EACH r.l[i] = a.l[i]; return r; }
template <int K> ROW_FN rw row_ror(const rw& a) { rw r; ROW_EACH r.l[i] = a.l[(i & 48) | ((i - K) & 15)]; return r; }          // lane l takes lane l - K of its row, cyclically
template <int K> ROW_FN rw row_bcast(const rw& a) { rw r; ROW_EACH r.l[i] = a.l[(i & 48) | K]; return r; }
ROW_FN rw row_up1(const rw& a) { rw r; ROW_EACH r.l[i] = (i & 15) ? a.l[i - 1] : 0; return r; }                                   // lane l takes lane l - 1; lane 0 takes 0
ROW_FN rw row_down1(const rw& a) { rw r; ROW_EACH r.l[i] = (i & 15) != 15 ? a.l[i + 1] : 0; return r; }                           // lane l takes lane l + 1
template <int R> ROW_FN rw row_from(const rw& a) { rw r; ROW_EACH r.l[i] = a.l[R * 16 + (i & 15)]; return r; }                    // every row takes row R's value
ROW_FN rw row_pick(const rw& a0, const rw& a1, const rw& a2, const rw& a3) { rw r; ROW_EACH r.l[i] = (i >> 4) == 0 ? a0.l[i] : (i >> 4) == 1 ? a1.l[i] : (i >> 4) == 2 ? a2.l[i] : a3.l[i]; return r; }
ROW_FN rw row_lanes(const int32_t (&t)[16]) { rw r; ROW_EACH r.l[i] = t[i & 15]; return r; }
template <int K> ROW_FN rw row_xrow(const rw& a) { rw r; ROW_EACH r.l[i] = a.l[i ^ (16 * K)]; return r; }                         // the value of row (r xor K) of the wave
ROW_FN rw row_sbyte(const rw& w, int k) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)w.l[i] << (24 - 8 * k)) >> 24; return r; }
ROW_FN rw row_lo16(const rw& w) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)w.l[i] & 0xffffu); return r; }
ROW_FN rw row_hi16(const rw& w) { rw r; ROW_EACH r.l[i] = (int32_t)((uint32_t)w.l[i] >> 16); return r; }
ROW_FN rw row_splat(int32_t v) { rw r; ROW_EACH r.l[i] = v; return r; }
ROW_FN rw row_sar64(const rw64& a, int n) { rw r; ROW_EACH { int64_t v = a.l[i] >> n; BLS_REQUIRE(v == (int32_t)v, "row_sar64: beyond 32 bits"); r.l[i] = (int32_t)v; } return r; }
ROW_FN rw64 row_add64(const rw64& a, int64_t c) { rw64 r; ROW_EACH r.l[i] = a.l[i] + c; return r; }
ROW_FN rw row_neg(const rw& a) { rw r; ROW_EACH r.l[i] = -a.l[i]; return r; }
ROW_FN rw row_load(const uint32_t* w) { rw r; ROW_EACH r.l[i] = (i & 15) < FP_N ? (int32_t)w[i & 15] : 0; return r; }             // an Fp value stored as 14 words
ROW_FN void row_store(uint32_t* w, const rw& a) { for (int i = 0; i < FP_N; i++) w[i] = (uint32_t)a.l[i]; }                      // row 0's
ROW_FN rw row_from_fp(const fp& v) { rw r; for (int i = 0; i < 64; i++) r.l[i] = (i & 15) < FP_N ? (int32_t)v.l[i & 15] : 0; return r; }
ROW_FN fp row_to_fp(const rw& a) {                                                                                              // row 0's value, as a one-lane Fp
    fp r;
    for (int i = 0; i < FP_N; i++) r.l[i] = (uint32_t)a.l[i];
    BLS_SET_VB(r, 64);
    BLS_SET_LB(r, 2);
    return r;
}
#undef ROW_EACH
#else
typedef int32_t rw;
typedef int64_t rw64;
#define ROW_FN __device__ __forceinline__
ROW_FN rw row_and(rw a, uint32_t m) { return (rw)((uint32_t)a & m); }
ROW_FN rw row_shl(rw a, int n) { return (rw)((uint32_t)a << n); }
ROW_FN rw row_sar(rw a, int n) { return a >> n; }
ROW_FN rw row_sext2(rw a) { return (rw)((uint32_t)a << 30) >> 30; }
ROW_FN rw row_mullo(rw a, uint32_t c) { return (rw)((uint32_t)a * c); }
ROW_FN rw64 row_mad(rw a, rw b, rw64 acc) { return bls_mac(acc, a, b); }
ROW_FN rw64 row_zero64() { return 0; }
ROW_FN rw row_zero() { return 0; }
ROW_FN rw row_lo(rw64 a) { return (rw)(uint32_t)a; }
ROW_FN rw row_hi28(rw64 a) { return (rw)(uint32_t)((uint64_t)a >> 28); }                       // v_alignbit_b32
ROW_FN rw64 row_ext(rw a) { return (rw64)a; }
// DPP moves with every lane written (bound_ctrl, full masks): the old value is dead, no initialising move
template <int K> ROW_FN rw row_ror(rw a) { return K == 0 ? a : __builtin_amdgcn_update_dpp(0, a, 0x120 + (K ? K : 1), 0xf, 0xf, true); }
template <int K> ROW_FN rw row_bcast(rw a) { return __builtin_amdgcn_update_dpp(0, a, 0x150 + K, 0xf, 0xf, true); }        // row_newbcast (gfx90a and later)
ROW_FN rw row_up1(rw a) { return __builtin_amdgcn_update_dpp(0, a, 0x111, 0xf, 0xf, true); }                               // row_shr:1, zero into lane 0
ROW_FN rw row_down1(rw a) { return __builtin_amdgcn_update_dpp(0, a, 0x101, 0xf, 0xf, true); }                             // row_shl:1, zero into lane 15
template <int R> ROW_FN rw row_from(rw a) { return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 15u) | (R * 16u)) << 2), a); }
ROW_FN rw row_pick(rw a0, rw a1, rw a2, rw a3) {
    const uint32_t r = (threadIdx.x >> 4) & 3u;
    return r == 0 ? a0 : (r == 1 ? a1 : (r == 2 ? a2 : a3));
}
template <int K> ROW_FN rw row_xrow(rw a) { return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u) ^ (16u * K)) << 2), a); }
ROW_FN rw row_sbyte(rw w, int k) { return (rw)((uint32_t)w << (24 - 8 * k)) >> 24; }
ROW_FN rw row_lo16(rw w) { return (rw)((uint32_t)w & 0xffffu); }
ROW_FN rw row_hi16(rw w) { return (rw)((uint32_t)w >> 16); }
ROW_FN rw row_splat(int32_t v) { return v; }
ROW_FN rw row_sar64(rw64 a, int n) { return (rw)(a >> n); }
ROW_FN rw64 row_add64(rw64 a, int64_t c) { return a + c; }
ROW_FN rw row_neg(rw a) { return -a; }
ROW_FN rw row_lanes(const int32_t (&t)[16]) {
    const uint32_t l = threadIdx.x & 15u;
    rw r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r = l == (uint32_t)i ? t[i] : r;
    return r;
}
ROW_FN rw row_load(const uint32_t* w) {
    const uint32_t l = threadIdx.x & 15u;
    return l < (uint32_t)FP_N ? (rw)w[l] : 0;
}
ROW_FN void row_store(uint32_t* w, rw a) {
    if ((threadIdx.x & 63u) < (uint32_t)FP_N) w[threadIdx.x & 63u] = (uint32_t)a;
}
ROW_FN rw row_from_fp(const fp& v) {                     // a one-lane Fp that every lane holds -> its limbs along the rows
    const uint32_t l = threadIdx.x & 15u;
    rw r = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r = l == (uint32_t)i ? (rw)v.l[i] : r;
    return r;
}
ROW_FN fp row_to_fp(rw a) {                              // row 0's value in every lane (wave-uniform: 14 v_readlane)
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = (uint32_t)__builtin_amdgcn_readlane(a, i);
    return r;
}
#endif

// lane constants of a row: p rotated for the fourteen steps, the masks of the linear carry pass
struct row_ctx {
    rw prot[FP_N];        // lane l: p_((l - k) mod 16), p_14 = p_15 = 0
    rw maskv;             // 2^28 - 1 in lanes 0..12, all ones in lane 13 (the top limb keeps its sign), 0 in lanes 14, 15
    rw nmaskv;            // the bits above 2^28 of lanes 0..12, nothing elsewhere
    rw lane13;            // all ones in lane 13
    rw live;              // all ones in lanes 0..13
    rw low13;             // all ones in lanes 0..12 (the limbs that pass a carry on)
};
template <int K>
ROW_FN void row_ctx_fill(row_ctx& C, const rw& p) {
    C.prot[K] = row_ror<K>(p);
    if constexpr (K + 1 < FP_N) row_ctx_fill<K + 1>(C, p);
}
ROW_FN row_ctx row_ctx_make() {
    int32_t P16[16], M[16], NM[16], L13[16], LV[16], LO[16];
    for (int i = 0; i < 16; i++) {
        P16[i] = i < FP_N ? (int32_t)k::P[i] : 0;
        M[i] = i < FP_N - 1 ? (int32_t)FP_MASK : (i == FP_N - 1 ? -1 : 0);
        NM[i] = i < FP_N - 1 ? (int32_t)~FP_MASK : 0;
        L13[i] = i == FP_N - 1 ? -1 : 0;
        LV[i] = i < FP_N ? -1 : 0;
        LO[i] = i < FP_N - 1 ? -1 : 0;
    }
    row_ctx C;
    row_ctx_fill<0>(C, row_lanes(P16));
    C.maskv = row_lanes(M);
    C.nmaskv = row_lanes(NM);
    C.lane13 = row_lanes(L13);
    C.live = row_lanes(LV);
    C.low13 = row_lanes(LO);
    return C;
}

// (the operands of step K + 1 - a broadcast and a rotation that depend on nothing in the chain - are formed right behind step K's first multiply-add, where the
// DPP read of the fresh accumulator would otherwise wait two states)
template <int K>
ROW_FN void row_mul_step(const row_ctx& C, const rw& a, const rw& b, const rw& ak, const rw& bk, rw64& acc) {
    acc = row_mad(ak, bk, acc);                                                   // column l (or l + 16) += a_K * b_(l - K)
    constexpr int K1 = K + 1 < FP_N ? K + 1 : K;
    const rw ak1 = row_bcast<K1>(a), bk1 = row_ror<K1>(b);
    const rw m = row_and(row_mullo(row_bcast<K>(row_lo(acc)), k::N0), FP_MASK);   // column K's Montgomery digit, in every lane
    acc = row_mad(m, C.prot[K], acc);                                             // column K is now a multiple of 2^28
    const rw hi = row_hi28(acc);                                                  // (first: the mask below then fills the wait states of the DPP read of `hi`)
    const rw lo = row_and(row_lo(acc), FP_MASK);
    acc = row_ext(lo + row_ror<1>(hi));                                           // every lane keeps 28 bits and takes its neighbour's rest; lane K restarts as column K + 16
    if constexpr (K + 1 < FP_N) row_mul_step<K + 1>(C, a, b, ak1, bk1, acc);
}
// a * b / 2^392 mod p.  In: limbs |.| <= 2^29 + 64, lanes 14 and 15 zero.  Out: limbs 0..12 in [0, 2^28) plus a carry of at most 16 in size, limb 13 signed,
// lanes 14, 15 zero; the value is a b / 2^392 + (0 .. 1) p.
ROW_FN rw row_mul(const row_ctx& C, const rw& a, const rw& b) {
    rw64 acc = row_zero64();
    row_mul_step<0>(C, a, b, row_bcast<0>(a), row_ror<0>(b), acc);
    // lanes 0..13 hold columns 16..29, lanes 14, 15 columns 14, 15: limb j of the result is column 14 + j
    rw r = row_ror<2>(row_lo(acc));
    r = row_and(r, FP_MASK) + row_up1(row_sar(r, 28));                            // one linear carry pass over the sixteen limbs (what leaves limb 15 is sign extension)
    // limbs 14, 15 are now 0, or the sign extension of a negative value, or a carry waiting above limb 13: all of it is (limb 14 mod 4, signed) * 2^28 in limb 13
    const rw top = row_shl(row_sext2(row_down1(r)), 28) & C.lane13;
    return (r + top) & C.live;
}
// linear combinations between products: one carry pass (limbs |.| < 2^31 in, < 2^28 + 8 out; the top limb keeps everything above it)
ROW_FN rw row_norm(const row_ctx& C, const rw& x) { return (x & C.maskv) + row_up1(row_sar(x & C.nmaskv, 28)); }

struct row_g1 { rw x, y, z; };       // a Jacobian point, the same in the four rows of the wave

// 2 P (dbl-2009-l, a = 0): X^2 | Y^2 | Y Z, then B^2 | (X + B)^2 | E^2, then E (D - X3) in the rows of the wave.  Infinity (Z = 0) stays infinity; E(Fp) has
// no point of order two.  In and out: limbs as row_mul leaves them (Z: twice that), |values| below 32 p.
ROW_FN row_g1 row_dbl(const row_ctx& C, const row_g1& p) {
    const rw p1 = row_mul(C, row_pick(p.x, p.y, p.y, p.y), row_pick(p.x, p.y, p.z, p.z));
    const rw A = row_from<0>(p1), B = row_from<1>(p1), YZ = row_from<2>(p1);
    const rw E = row_norm(C, A + A + A);
    const rw s2 = row_pick(B, p.x + B, E, E);
    const rw p2 = row_mul(C, s2, s2);
    const rw Cq = row_from<0>(p2), t = row_from<1>(p2), Fq = row_from<2>(p2);
    const rw D1 = t - A - Cq;
    const rw D = row_norm(C, D1 + D1);
    row_g1 r;
    r.x = row_norm(C, Fq - D - D);
    const rw C4 = row_norm(C, row_shl(Cq, 2));
    r.y = row_norm(C, row_mul(C, E, row_norm(C, D - r.x)) - C4 - C4);
    r.z = YZ + YZ;
    return r;
}

// the rare branches of an addition are decided on one-lane values (any representative of a residue: fp_is_zero reduces first)
ROW_FN bool row_is_zero(const rw& a) { return fp_is_zero(row_to_fp(a)); }

// P + Q, complete: five rounds of products (Z1^2 | Z2^2;  X1 Z2Z2 | X2 Z1Z1 | Y1 Z2 | Y2 Z1;  T1 Z2Z2 | T2 Z1Z1 | Z1 Z2 | H^2;  H HH | U1 HH | Z12 H | r^2;
// r (V - X3) | S1 HHH).  Every exceptional case (an operand at infinity, P = Q, P = -Q) ends in Z3 = 0 mod p: only then are the operands looked at.
ROW_FN row_g1 row_add(const row_ctx& C, const row_g1& p, const row_g1& q) {
    const rw zsel = row_pick(p.z, q.z, p.z, q.z);
    const rw p1 = row_mul(C, zsel, zsel);
    const rw Z1Z1 = row_from<0>(p1), Z2Z2 = row_from<1>(p1);
    const rw p2 = row_mul(C, row_pick(p.x, q.x, p.y, q.y), row_pick(Z2Z2, Z1Z1, q.z, p.z));
    const rw U1 = row_from<0>(p2), U2 = row_from<1>(p2), T1 = row_from<2>(p2), T2 = row_from<3>(p2);
    const rw H = U2 - U1;
    const rw p3 = row_mul(C, row_pick(T1, T2, p.z, H), row_pick(Z2Z2, Z1Z1, q.z, H));
    const rw S1 = row_from<0>(p3), S2 = row_from<1>(p3), Z12 = row_from<2>(p3), HH = row_from<3>(p3);
    const rw rr = S2 - S1;
    const rw p4 = row_mul(C, row_pick(H, U1, Z12, rr), row_pick(HH, HH, H, rr));
    const rw HHH = row_from<0>(p4), V = row_from<1>(p4), Z3 = row_from<2>(p4), RR = row_from<3>(p4);
    row_g1 r;
    r.x = row_norm(C, RR - HHH - V - V);
    const rw p5 = row_mul(C, row_pick(rr, S1, rr, S1), row_pick(row_norm(C, V - r.x), HHH, row_norm(C, V - r.x), HHH));
    r.y = row_norm(C, row_from<0>(p5) - row_from<1>(p5));
    r.z = Z3;
    if (row_is_zero(Z3)) {
        const bool pinf = row_is_zero(p.z), qinf = row_is_zero(q.z);
        if (pinf) return q;
        if (qinf) return p;
        if (row_is_zero(H) && row_is_zero(rr)) return row_dbl(C, p);
        // P = -Q: Z3 = 0 is the answer
    }
    return r;
}

// a^e along a row for a fixed exponent given as fp_pow_sched's 5-bit sliding-window schedule (constants.hpp: pairs of (squarings, odd multiplier)).  The table of
// odd powers is kept by `tab` (put / get by index: LDS on the device - one word per lane and entry - an array on the CPU).  In: limbs as row_mul takes them.
template <class Tab>
ROW_FN rw row_pow_sched(const row_ctx& C, const rw& a, const uint8_t (*sched)[2], int len, Tab& tab) {
    tab.put(0, a);
    const rw a2 = row_mul(C, a, a);
    rw t = a;
#pragma clang loop unroll(disable)
    for (int i = 1; i < 16; i++) {
        t = row_mul(C, t, a2);
        tab.put(i, t);
    }
    rw r = tab.get(sched[0][1] >> 1);
#pragma clang loop unroll(disable)
    for (int j = 1; j < len; j++) {
        const int nsq = sched[j][0];
#pragma clang loop unroll(disable)
        for (int i = 0; i < nsq; i++) r = row_mul(C, r, r);
        const uint32_t v = sched[j][1];
        if (v) r = row_mul(C, r, tab.get(v >> 1));
    }
    return r;
}
#if defined(BLS_ROW_EMU)
struct row_tab_array {
    rw t[16];
    void put(int i, const rw& v) { t[i] = v; }
    rw get(int i) const { return t[i]; }
};
#else
struct row_tab_lds {                   // 16 entries x the block's lanes, one word each
    uint32_t* base;                    // LDS, this lane's column: entry i at base[i * stride]
    uint32_t stride;
    __device__ __forceinline__ void put(int i, rw v) { base[i * stride] = (uint32_t)v; }
    __device__ __forceinline__ rw get(int i) const { return (rw)base[i * stride]; }
};
#endif

}  // namespace bls
