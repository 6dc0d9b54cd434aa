// Fp = GF(p), BLS12-381 base field, for gfx950 lanes.
//
// Representation: 14 limbs of 28 bits in 32-bit words, Montgomery form with R = 2^392.
// Why not the reference's 12 x 32 / 6 x 64 saturated limbs (blst_fp, blst_abi.nim:87-94)?  CDNA4's
// integer multiplier is v_mad_u64_u32 (32x32 + 64 -> 64).  It has a carry-OUT but no carry-IN, so a
// saturated multi-precision MAC needs a second instruction (v_addc_co_u32) per partial product.  With
// 28-bit limbs a whole Montgomery column -- up to 14 a_i*b_j plus 14 m_i*p_j products of < 2^60 -- sums
// in one 64-bit accumulator without overflow: one instruction per partial product (392 per field
// multiplication), no carry chains, and the compiler emits exactly that from plain C++.
// MFMA is not used: this is per-lane multi-precision arithmetic, not a dense contraction.
//
// Limbs are SIGNED (two's complement in the 32-bit word; products use v_mad_i64_i32) and values are
// "semi-normalised": limbs 0..12 in [-4, 2^28 + 4), limb 13 holds the signed excess, and the VALUE is only
// bounded in magnitude by a small multiple of p (never reduced into [0, p) except by fp_canon).
// Additions and subtractions are limb-wise plus ONE parallel carry step (no sequential carry chain, no
// conditional subtraction, no bias constant), so value bounds grow additively.  fp_mul accepts limbs of
// magnitude < 2^29 + 2^20 (a carry-less sum of two elements) and value bounds whose product is <= 2048 p^2, and returns a value in (-2p, 2p).
// The host test build (-DBLS_TRACK_BOUNDS, tests/host_emu) carries a worst-case magnitude bound with
// every element and asserts these preconditions on every operation, independent of the data.
//
// The reference's memory image (12 x 32-bit limbs, R = 2^384) is converted at the boundary only
// (fp_from_blst / fp_to_blst: one multiplication each).
//
// All functions are __host__ __device__ so that tests/host_emu can execute the exact kernel
// arithmetic on the build container's CPU (there is no GPU there); the product only ever calls
// them from __global__ kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BLS_HD __host__ __device__ __forceinline__
#define BLS_HDN __host__ __device__ __noinline__
// Mid-level formulas: inlined on the device so operands stay in VGPRs/AGPRs between the register-argument
// calls into the shared multiplier body; kept out-of-line on the host to keep tests/host_emu quick to build.
#if defined(__HIP_DEVICE_COMPILE__)
#define BLS_MID __device__ __forceinline__
#else
#define BLS_MID __host__ __device__ __noinline__
#endif
#define BLS_CONST static constexpr
#include "constants.hpp"

#if defined(BLS_TRACK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#include <execinfo.h>
#define BLS_VB_FIELD uint32_t vb, lb;
// multiply-add census of the host test build (tests/host_emu): every multiplier body adds its v_mad_i64_i32 count, so the
// op-count model behind bench.py's roofline.int_mad is measured from the real formulas, not estimated
namespace bls { inline unsigned long long g_mad_count = 0; }
#define BLS_COUNT_MADS(n) (::bls::g_mad_count += (n))
#define BLS_SET_VB(x, v) ((x).vb = (v))
#define BLS_VB(x) ((x).vb)
#define BLS_SET_LB(x, v) ((x).lb = (v))
#define BLS_LB(x) ((x).lb)
#define BLS_REQUIRE(cond, what)                                                         \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            std::fprintf(stderr, "fp bound violation: %s (%s:%d)\n", what, __FILE__, __LINE__); \
            void* bt_[24];                                                              \
            backtrace_symbols_fd(bt_, backtrace(bt_, 24), 2);                           \
            std::abort();                                                               \
        }                                                                               \
    } while (0)
#else
#define BLS_VB_FIELD
#define BLS_COUNT_MADS(n) ((void)0)
#define BLS_SET_VB(x, v) ((void)0)
#define BLS_VB(x) 0u
#define BLS_SET_LB(x, v) ((void)0)
#define BLS_LB(x) 0u
#define BLS_REQUIRE(cond, what) ((void)0)
#endif

namespace bls {

constexpr int FP_N = 14;
constexpr uint32_t FP_MASK = 0x0fffffffu;

struct fp {
    uint32_t l[FP_N];
    BLS_VB_FIELD
};

BLS_HD fp fp_from_const(const uint32_t (&c)[FP_N]) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = c[i];
    BLS_SET_VB(r, 1);
    BLS_SET_LB(r, 0);        // the generated constants are canonical: limbs 0..12 in [0, 2^28)
    return r;
}

BLS_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = 0;
    BLS_SET_VB(r, 1);
    BLS_SET_LB(r, 0);
    return r;
}

BLS_HD fp fp_one() { return fp_from_const(k::ONE); }

// r = c ? a : b
BLS_HD fp fp_select(bool c, const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = c ? a.l[i] : b.l[i];
    BLS_SET_VB(r, BLS_VB(a) > BLS_VB(b) ? BLS_VB(a) : BLS_VB(b));
    BLS_SET_LB(r, BLS_LB(a) > BLS_LB(b) ? BLS_LB(a) : BLS_LB(b));
    return r;
}

// one parallel carry step (arithmetic shifts: limbs may be slightly negative): |limbs| < 2^31 in,
// limbs 0..12 in [-4, 2^28 + 4) out, limb 13 absorbs the rest
BLS_HD void fp_carry_step(uint32_t (&s)[FP_N]) {
    uint32_t c[FP_N - 1];
#pragma unroll
    for (int i = 0; i < FP_N - 1; i++) c[i] = (uint32_t)((int32_t)s[i] >> 28);
    s[0] &= FP_MASK;
#pragma unroll
    for (int i = 1; i < FP_N - 1; i++) s[i] = (s[i] & FP_MASK) + c[i - 1];
    s[FP_N - 1] += c[FP_N - 2];
}

// Limb-magnitude bookkeeping (host tracker): lb = worst-case |limb| in units of 2^28.  Carried values have
// lb 1; limb-wise ("_nc") operations add the operands' lb; int32 limbs hold at most 7 units; fp_mul / fp_sqr
// accept lb <= 2.
BLS_HD fp fp_add_nc(const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = a.l[i] + b.l[i];
    BLS_SET_VB(r, BLS_VB(a) + BLS_VB(b));
    BLS_SET_LB(r, (BLS_LB(a) ? BLS_LB(a) : 1) + (BLS_LB(b) ? BLS_LB(b) : 1));
    BLS_REQUIRE(BLS_LB(r) <= 7 && BLS_VB(r) <= 1024, "fp_add_nc bounds");
    return r;
}
BLS_HD fp fp_sub_nc(const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = a.l[i] - b.l[i];
    BLS_SET_VB(r, BLS_VB(a) + BLS_VB(b));
    BLS_SET_LB(r, (BLS_LB(a) ? BLS_LB(a) : 1) + (BLS_LB(b) ? BLS_LB(b) : 1));
    BLS_REQUIRE(BLS_LB(r) <= 7 && BLS_VB(r) <= 1024, "fp_sub_nc bounds");
    return r;
}
// difference of two values whose limbs 0..12 are NON-NEGATIVE and below 2^28 (fresh multiplication results,
// fp_reduce / fp_carry_full results): |limb| stays below 2^28, no carry needed
BLS_HD fp fp_sub_pos(const fp& a, const fp& b) {
    BLS_REQUIRE(BLS_LB(a) == 0 && BLS_LB(b) == 0, "fp_sub_pos needs canonical-limb operands");
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = a.l[i] - b.l[i];
    BLS_SET_VB(r, BLS_VB(a) + BLS_VB(b));
    BLS_SET_LB(r, 1);
    return r;
}
// carry step on a limb-wise result: back to lb 1
BLS_HD fp fp_carry(const fp& a) {
    BLS_REQUIRE(BLS_LB(a) <= 7, "fp_carry limb bound");
    fp r = a;
    fp_carry_step(r.l);
    BLS_SET_LB(r, 1);
    return r;
}
BLS_HD fp fp_add(const fp& a, const fp& b) { return fp_carry(fp_add_nc(a, b)); }
BLS_HD fp fp_sub(const fp& a, const fp& b) { return fp_carry(fp_sub_nc(a, b)); }
BLS_HD fp fp_dbl_nc(const fp& a) { return fp_add_nc(a, a); }

// limb-wise negation: magnitudes are unchanged, so no carry is needed
BLS_HD fp fp_neg(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = 0u - a.l[i];
    BLS_SET_VB(r, BLS_VB(a));
    BLS_SET_LB(r, BLS_LB(a) ? BLS_LB(a) : 1);
    return r;
}

BLS_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }

// acc + a * b as ONE v_mad_i64_i32 whose addend IS the running accumulator.  Written as an asm statement on the device because
// hipcc otherwise re-associates the column sum: it starts every column in a second register pair from 0 and joins the two with
// an extra 64-bit add (v_lshl_add_u64) per column - 26 instructions of the ~490 of a product for instruction-level parallelism
// that this stream does not need (a dependent chain of multiply-adds issues as fast as an independent one, tools/ubench_valu.hip).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int64_t bls_mac(int64_t acc, int32_t a, int32_t b) {
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    return acc;
}
__device__ __forceinline__ int64_t bls_mac_c(int64_t acc, int32_t a, uint32_t c) {      // c: a limb of p (scalar register)
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(c) : "vcc");
    return acc;
}
#else
BLS_HD int64_t bls_mac(int64_t acc, int32_t a, int32_t b) { return acc + (int64_t)a * b; }
BLS_HD int64_t bls_mac_c(int64_t acc, int32_t a, uint32_t c) { return acc + (int64_t)a * (int32_t)c; }
#endif

// Montgomery product a*b*2^-392 mod p: product scanning, ONE signed 64-bit accumulator, 392 multiply-adds
// (v_mad_i64_i32).  Operand limbs are at most limb-wise sums of two semi-normalised values (|l| < 2^29 + 16):
// column bound 14 * (2^29.01)^2 + 14 * 2^56 + carry < 2^62.  m_k in [0, 2^28) makes the low 28
// bits of the column vanish; the arithmetic shift is then an exact division.  Result in (-2p, 2p).
BLS_HD fp fp_mul_core(const fp& a, const fp& b) {
    int64_t acc = 0;
    int32_t m[FP_N];
    fp r;
#pragma unroll
    for (int kk = 0; kk < FP_N; kk++) {
#pragma unroll
        for (int i = 0; i <= kk; i++) acc = bls_mac(acc, (int32_t)a.l[i], (int32_t)b.l[kk - i]);
#pragma unroll
        for (int i = 0; i < kk; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        m[kk] = (int32_t)(((uint32_t)acc * k::N0) & FP_MASK);
        acc = bls_mac_c(acc, m[kk], k::P[0]);
        acc >>= 28;
    }
#pragma unroll
    for (int kk = FP_N; kk < 2 * FP_N - 1; kk++) {
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac(acc, (int32_t)a.l[i], (int32_t)b.l[kk - i]);
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        r.l[kk - FP_N] = (uint32_t)acc & FP_MASK;
        acc >>= 28;
    }
    r.l[FP_N - 1] = (uint32_t)acc;
    BLS_SET_VB(r, 2);
    BLS_SET_LB(r, 0);        // limbs 0..12 non-negative and < 2^28
    return r;
}

// Montgomery square: the 91 off-diagonal products are formed once against the doubled operand, so the
// operand half costs 105 multiply-adds instead of 196 (301 in all).  Same bounds as fp_mul_core.
BLS_HD fp fp_sqr_core(const fp& a) {
    int64_t acc = 0;
    int32_t m[FP_N], a2[FP_N];
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) a2[i] = (int32_t)a.l[i] * 2;
#pragma unroll
    for (int kk = 0; kk < 2 * FP_N - 1; kk++) {
#pragma unroll
        for (int i = (kk < FP_N ? 0 : kk - FP_N + 1); 2 * i < kk; i++) acc = bls_mac(acc, a2[i], (int32_t)a.l[kk - i]);
        if ((kk & 1) == 0) acc = bls_mac(acc, (int32_t)a.l[kk / 2], (int32_t)a.l[kk / 2]);
        if (kk < FP_N) {
#pragma unroll
            for (int i = 0; i < kk; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
            m[kk] = (int32_t)(((uint32_t)acc * k::N0) & FP_MASK);
            acc = bls_mac_c(acc, m[kk], k::P[0]);
        } else {
#pragma unroll
            for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
            r.l[kk - FP_N] = (uint32_t)acc & FP_MASK;
        }
        acc >>= 28;
    }
    r.l[FP_N - 1] = (uint32_t)acc;
    BLS_SET_VB(r, 2);
    BLS_SET_LB(r, 0);
    return r;
}


// Montgomery "dot product" (a*b + c*d) * 2^-392 mod p with ONE reduction: 392 operand multiply-adds + 196 for the
// reduction instead of 2 x 392 (lazy reduction; an Fp2 product is two of these and needs no Karatsuba
// additions, subtractions or carries).  Column bound with limbs < 2^29 + 2^20: 28 * 2^58.01 + 14 * 2^56 < 2^63.
BLS_HD fp fp_dot2_core(const fp& a, const fp& b, const fp& c, const fp& d) {
    int64_t acc = 0;
    int32_t m[FP_N];
    fp r;
#pragma unroll
    for (int kk = 0; kk < FP_N; kk++) {
#pragma unroll
        for (int i = 0; i <= kk; i++) acc = bls_mac(acc, (int32_t)a.l[i], (int32_t)b.l[kk - i]);
#pragma unroll
        for (int i = 0; i <= kk; i++) acc = bls_mac(acc, (int32_t)c.l[i], (int32_t)d.l[kk - i]);
#pragma unroll
        for (int i = 0; i < kk; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        m[kk] = (int32_t)(((uint32_t)acc * k::N0) & FP_MASK);
        acc = bls_mac_c(acc, m[kk], k::P[0]);
        acc >>= 28;
    }
#pragma unroll
    for (int kk = FP_N; kk < 2 * FP_N - 1; kk++) {
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac(acc, (int32_t)a.l[i], (int32_t)b.l[kk - i]);
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac(acc, (int32_t)c.l[i], (int32_t)d.l[kk - i]);
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        r.l[kk - FP_N] = (uint32_t)acc & FP_MASK;
        acc >>= 28;
    }
    r.l[FP_N - 1] = (uint32_t)acc;
    BLS_SET_VB(r, 2);
    BLS_SET_LB(r, 0);
    return r;
}

// Montgomery dot product (sum_t x_t * y_t) * 2^-392 mod p of N pairs with ONE reduction: 196 N operand multiply-adds + 196 for the
// reduction (the general form of fp_dot2_core).  What a sum of products costs in THIS machine is its instruction count (every VALU
// instruction takes the same issue slot at one wave per SIMD), and a reduction is 196 multiply-adds + 68 bookkeeping instructions
// plus whatever additions, subtractions and carry steps combine separately reduced products afterwards: a schoolbook sum with one
// reduction per output coefficient beats a Karatsuba tree of separately reduced products wherever the tree's glue is heavy
// (fp12_mul_by_line_lazy: 12 of these instead of 13 Fp2 products, 26 reductions and ~1 500 instructions of glue).
// Column bound: sum_t |x_t limb| |y_t limb| * 14 + 14 * 2^56 + carry < 2^63, i.e. sum_t lb(x_t) lb(y_t) <= 8 limb units (2^28 each,
// with the usual 2^20 of slack per limb); value bound: sum_t vb(x_t) vb(y_t) <= 2048.  Result in (-2p, 2p), canonical limbs.
template <int N>
BLS_HD fp fp_dotn_core(const fp (&x)[N], const fp (&y)[N]) {
    int64_t acc = 0;
    int32_t m[FP_N];
    fp r;
#pragma unroll
    for (int kk = 0; kk < FP_N; kk++) {
#pragma unroll
        for (int t = 0; t < N; t++) {
#pragma unroll
            for (int i = 0; i <= kk; i++) acc = bls_mac(acc, (int32_t)x[t].l[i], (int32_t)y[t].l[kk - i]);
        }
#pragma unroll
        for (int i = 0; i < kk; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        m[kk] = (int32_t)(((uint32_t)acc * k::N0) & FP_MASK);
        acc = bls_mac_c(acc, m[kk], k::P[0]);
        acc >>= 28;
    }
#pragma unroll
    for (int kk = FP_N; kk < 2 * FP_N - 1; kk++) {
#pragma unroll
        for (int t = 0; t < N; t++) {
#pragma unroll
            for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac(acc, (int32_t)x[t].l[i], (int32_t)y[t].l[kk - i]);
        }
#pragma unroll
        for (int i = kk - FP_N + 1; i < FP_N; i++) acc = bls_mac_c(acc, m[i], k::P[kk - i]);
        r.l[kk - FP_N] = (uint32_t)acc & FP_MASK;
        acc >>= 28;
    }
    r.l[FP_N - 1] = (uint32_t)acc;
    BLS_SET_VB(r, 2);
    BLS_SET_LB(r, 0);
    return r;
}
// checked form (host: bounds tracker + census; device: the core in place)
template <int N>
BLS_HD fp fp_dotn(const fp (&x)[N], const fp (&y)[N]) {
#if defined(BLS_TRACK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
    uint64_t vsum = 0, lsum = 0;
    const int64_t LIM1 = (1ll << 28) + (1ll << 20);
    for (int t = 0; t < N; t++) {
        vsum += (uint64_t)BLS_VB(x[t]) * BLS_VB(y[t]);
        int64_t mx = 0, my = 0;                       // measured magnitudes must fit the DECLARED limb units
        for (int i = 0; i < FP_N; i++) {
            int64_t a = (int32_t)x[t].l[i], b = (int32_t)y[t].l[i];
            if (a < 0) a = -a;
            if (b < 0) b = -b;
            if (a > mx) mx = a;
            if (b > my) my = b;
        }
        uint64_t lx = BLS_LB(x[t]) ? BLS_LB(x[t]) : 1, ly = BLS_LB(y[t]) ? BLS_LB(y[t]) : 1;
        BLS_REQUIRE(mx < (int64_t)lx * LIM1 && my < (int64_t)ly * LIM1, "fp_dotn limb magnitude exceeds its declared units");
        lsum += lx * ly;
    }
    BLS_REQUIRE(vsum <= 2048, "fp_dotn value bounds");
    BLS_REQUIRE(lsum <= 8, "fp_dotn limb-unit bounds (column sum)");
#endif
    BLS_COUNT_MADS(196 * N + 196);
    return fp_dotn_core<N>(x, y);
}

#if defined(__HIP_DEVICE_COMPILE__)
// The multiplier body (~500 instructions) is shared by every caller so that hot loops fit the 64 KB
// instruction cache.  Operands travel in VGPRs: 28 scalar parameters map to v0..v27 and the result
// returns in v0..v13 (aggregate by-reference parameters would go through scratch memory, which at
// 65 536 lanes no longer fits L2).
__device__ __noinline__ fp fp_mul_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                       uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t b0, uint32_t b1,
                                       uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9,
                                       uint32_t b10, uint32_t b11, uint32_t b12, uint32_t b13) {
    fp a{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}}, b{{b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13}};
    return fp_mul_core(a, b);
}
__device__ __noinline__ fp fp_sqr_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                       uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13) {
    fp a{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}};
    return fp_sqr_core(a);
}
// n squarings in one call (n >= 1): the exponentiations square 4 times per window, and a call costs the argument
// moves plus an instruction-fetch bubble at the jump and at the return
__device__ __noinline__ fp fp_sqr_n_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                         uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t n) {
    fp a{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}};
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < n; i++) a = fp_sqr_core(a);
    return a;
}
__device__ __forceinline__ fp fp_sqr_n(const fp& a, uint32_t n) {
    return fp_sqr_n_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13], n);
}
__device__ __forceinline__ fp fp_sqr(const fp& a) {
    return fp_sqr_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13]);
}
__device__ __forceinline__ fp fp_mul(const fp& a, const fp& b) {
    return fp_mul_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], a.l[12], a.l[13],
                       b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11], b.l[12], b.l[13]);
}
#else
__host__ __device__ __noinline__ inline fp fp_mul(const fp& a, const fp& b) {
#if defined(BLS_TRACK_BOUNDS)
    BLS_REQUIRE((uint64_t)BLS_VB(a) * BLS_VB(b) <= 2048, "fp_mul value bounds");
    BLS_REQUIRE(BLS_LB(a) <= 2 && BLS_LB(b) <= 2, "fp_mul limb-unit bounds");
    const int64_t LIM = (1ll << 29) + (1ll << 20);     // limb-wise sum of two semi-normalised values, not three
    for (int i = 0; i < FP_N; i++) {
        int64_t x = (int32_t)a.l[i], y = (int32_t)b.l[i];
        BLS_REQUIRE(x > -LIM && x < LIM && y > -LIM && y < LIM, "fp_mul limb bound");
    }
#endif
    BLS_COUNT_MADS(392);
    return fp_mul_core(a, b);
}
__host__ __device__ __noinline__ inline fp fp_sqr(const fp& a) {
#if defined(BLS_TRACK_BOUNDS)
    BLS_REQUIRE((uint64_t)BLS_VB(a) * BLS_VB(a) <= 2048, "fp_sqr value bounds");
    BLS_REQUIRE(BLS_LB(a) <= 2, "fp_sqr limb-unit bounds");
    for (int i = 0; i < FP_N; i++) {
        int64_t x = (int32_t)a.l[i];
        BLS_REQUIRE(x > -(1ll << 29) - (1ll << 20) && x < (1ll << 29) + (1ll << 20), "fp_sqr limb bound");
    }
#endif
    BLS_COUNT_MADS(301);
    return fp_sqr_core(a);
}
__host__ __device__ inline fp fp_sqr_n(const fp& a, uint32_t n) {
    fp r = a;
    for (uint32_t i = 0; i < n; i++) r = fp_sqr(r);
    return r;
}
__host__ __device__ __noinline__ inline fp fp_dot2(const fp& a, const fp& b, const fp& c, const fp& d) {
#if defined(BLS_TRACK_BOUNDS)
    BLS_REQUIRE((uint64_t)BLS_VB(a) * BLS_VB(b) + (uint64_t)BLS_VB(c) * BLS_VB(d) <= 2048, "fp_dot2 value bounds");
    BLS_REQUIRE(BLS_LB(a) <= 2 && BLS_LB(b) <= 2 && BLS_LB(c) <= 2 && BLS_LB(d) <= 2, "fp_dot2 limb-unit bounds");
    const int64_t LIM = (1ll << 29) + (1ll << 20);
    for (int i = 0; i < FP_N; i++) {
        int64_t x = (int32_t)a.l[i], y = (int32_t)b.l[i], z = (int32_t)c.l[i], w = (int32_t)d.l[i];
        BLS_REQUIRE(x > -LIM && x < LIM && y > -LIM && y < LIM && z > -LIM && z < LIM && w > -LIM && w < LIM, "fp_dot2 limb bound");
    }
#endif
    BLS_COUNT_MADS(588);
    return fp_dot2_core(a, b, c, d);
}
#endif

// a * small constant via additions
BLS_HD fp fp_mul3(const fp& a) { return fp_add(fp_dbl(a), a); }

// fully carried limbs (0..12 in [0, 2^28), limb 13 signed); the value is unchanged
BLS_HD fp fp_carry_full(const fp& a) {
    fp r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N - 1; i++) {
        uint32_t v = a.l[i] + c;
        r.l[i] = v & FP_MASK;
        c = (uint32_t)((int32_t)v >> 28);
    }
    r.l[FP_N - 1] = a.l[FP_N - 1] + c;
    BLS_SET_VB(r, BLS_VB(a));
    BLS_SET_LB(r, 0);
    return r;
}

// Partial reduction: subtracts round(a / p) * p, estimated from the top limb (a / 2^364 against p / 2^364 =
// 106513.18; reciprocal 2^40 / that).  Any |a| <= 1024p in, |r| < 0.51p out, limbs fully carried.
// ~50 instructions: the cheap way to stop value bounds from growing through chains of additions.
BLS_HD fp fp_reduce(const fp& a) {
    BLS_REQUIRE(BLS_VB(a) <= 1024 && BLS_LB(a) <= 7, "fp_reduce bound");
    const int64_t RECIP = 10322735;                       // round(2^40 / (p / 2^364))
    int32_t top = (int32_t)a.l[FP_N - 1] + ((int32_t)a.l[FP_N - 2] >> 28);
    int32_t q = (int32_t)(((int64_t)top * RECIP + (1ll << 39)) >> 40);
    fp r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < FP_N - 1; i++) {
        acc += (int64_t)(int32_t)a.l[i] - (int64_t)q * (int32_t)k::P[i];
        r.l[i] = (uint32_t)acc & FP_MASK;
        acc >>= 28;
    }
    acc += (int64_t)(int32_t)a.l[FP_N - 1] - (int64_t)q * (int32_t)k::P[FP_N - 1];
    r.l[FP_N - 1] = (uint32_t)acc;
    BLS_SET_VB(r, 1);
    BLS_SET_LB(r, 0);
    return r;
}

// a == 0 mod p for any bounded a: reduce to |r| < p, then r must be exactly 0
BLS_HD bool fp_is_zero(const fp& a) {
    fp r = fp_reduce(a);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) acc |= r.l[i];
    return acc == 0;
}

BLS_HD bool fp_eq(const fp& a, const fp& b) { return fp_is_zero(fp_sub(a, b)); }

// the canonical representative in [0, p), fully carried
BLS_HD fp fp_canon(const fp& a) {
    fp r = fp_reduce(a);                                   // |r| < p, carried, limb 13 signed
    bool neg = (int32_t)r.l[FP_N - 1] < 0;
    fp u;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP_N - 1; i++) {
        uint32_t v = r.l[i] + (neg ? k::P[i] : 0u) + c;
        u.l[i] = v & FP_MASK;
        c = v >> 28;
    }
    u.l[FP_N - 1] = r.l[FP_N - 1] + (neg ? k::P[FP_N - 1] : 0u) + c;
    BLS_SET_VB(u, 1);
    BLS_SET_LB(u, 0);
    return u;
}

BLS_HD bool fp_limbs_are_zero(const fp& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) acc |= a.l[i];
    return acc == 0;
}

// Montgomery -> plain integer, canonical limbs
BLS_HD fp fp_from_mont(const fp& a) { return fp_canon(fp_mul(a, fp_from_const(k::PLAIN_ONE))); }

// plain integer < 2^392 given as 28-bit limbs (value bound set by the caller) -> Montgomery
BLS_HD fp fp_to_mont(const fp& a) { return fp_mul(a, fp_from_const(k::RR)); }

// 12 x 32-bit words (little-endian integer) <-> 14 x 28-bit limbs
BLS_HD fp fp_relimb_from32(const uint32_t (&w)[12]) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        const int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)w[wi] >> sh;
        if (sh > 4 && wi + 1 < 12) v |= (uint64_t)w[wi + 1] << (32 - sh);
        r.l[i] = (uint32_t)v & FP_MASK;
    }
    BLS_SET_VB(r, 16);      // any 384-bit integer is < 16p (2^384 / p < 10)
    BLS_SET_LB(r, 0);
    return r;
}
BLS_HD void fp_relimb_to32(uint32_t (&w)[12], const fp& a) {   // a fully carried, < 2^384
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, li = bit / 28, sh = bit % 28;
        uint64_t v = (uint64_t)a.l[li] >> sh;
        if (li + 1 < FP_N) v |= (uint64_t)a.l[li + 1] << (28 - sh);
        if (sh > 24 && li + 2 < FP_N) v |= (uint64_t)a.l[li + 2] << (56 - sh);
        w[j] = (uint32_t)v;
    }
}

// blst_fp memory image (Montgomery, R = 2^384) <-> device element (Montgomery, R = 2^392)
BLS_HD fp fp_from_blst(const uint32_t (&w)[12]) {
    fp v = fp_relimb_from32(w);
    return fp_mul(v, fp_from_const(k::C400));
}
BLS_HD void fp_to_blst(uint32_t (&w)[12], const fp& a) {
    fp r = fp_canon(fp_mul(a, fp_from_const(k::C384)));   // x * 2^384 mod p
    fp_relimb_to32(w, r);
}

// a^e for a 384-bit exponent given as 12 LE 32-bit words; fixed 4-bit window, not constant time
// (nothing secret on this path: public keys, messages, signatures and public blinding scalars).
// a^e for one of the two fixed exponents, by its 5-bit sliding-window schedule (constants.hpp: pairs of
// (squarings, odd multiplier)): 379 squarings + 67 multiplications + 16 for the table of odd powers.
BLS_HDN fp fp_pow_sched(const fp& a, const uint8_t (*sched)[2], int len) {
    fp tab[16];                                  // a^1, a^3, ..., a^31
    tab[0] = a;
    fp a2 = fp_sqr(a);
    for (int i = 1; i < 16; i++) tab[i] = fp_mul(tab[i - 1], a2);
    fp r = tab[sched[0][1] >> 1];
    for (int j = 1; j < len; j++) {
        r = fp_sqr_n(r, sched[j][0]);
        uint32_t v = sched[j][1];
        if (v) r = fp_mul(r, tab[v >> 1]);
    }
    return r;
}

// 1/a by Fermat (a^(p-2)): 379 squarings + 83 multiplications.  Reference for the tests of fp_inv below.
BLS_HD fp fp_inv_fermat(const fp& a) { return fp_pow_sched(a, k::SW_PM2, k::SW_PM2_LEN); }

// 1/a with the Bernstein-Yang division steps ("safegcd"), 62 at a time: the steps of a batch run on the low 64 bits of
// (f, g) and are collected in a 2x2 matrix t (|entries| <= 2^62), then (f, g) <- t (f, g) / 2^62 exactly and
// (d, e) <- t (d, e) / 2^62 mod p on 7 signed 62-bit limbs, until g = 0; then f = +-1 and d = +-1/x.  About 13 batches
// of ~4k instructions where the exponentiation takes ~190k: the one long single-lane step of the serial tail (the Fp
// inversion inside the final exponentiation's Fp12 inversion) drops from 0.43 to ~0.1 ms.  Not constant time (the loop ends
// when g = 0): nothing secret reaches an inversion on this path.  a = 0 -> 0, like the exponentiation.
struct inv62 {
    int64_t v[7];
};
BLS_HD void inv62_divsteps(int64_t& eta, uint64_t f, uint64_t g, int64_t (&t)[4]) {
    // eta = -delta.  Runs of even g are taken in one go (count-trailing-zeros under a sentinel at the steps left); an odd g
    // first swaps (f, g) <- (g, -f) when delta > 0, then g += f makes it even: about two steps per round.
    uint64_t u = 1, v = 0, q = 0, r = 1;
    int i = 62;
    for (;;) {
        const int zeros = __builtin_ctzll(g | (~0ull << i));
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {
            uint64_t tmp;
            eta = -eta;
            tmp = f; f = g; g = 0 - tmp;
            tmp = u; u = q; q = 0 - tmp;
            tmp = v; v = r; r = 0 - tmp;
        }
        g += f; q += u; r += v;
    }
    t[0] = (int64_t)u; t[1] = (int64_t)v; t[2] = (int64_t)q; t[3] = (int64_t)r;
}
BLS_HD void inv62_update_fg(inv62& f, inv62& g, const int64_t (&t)[4]) {
    const int64_t M62 = (int64_t)((1ull << 62) - 1);
    __int128 cf = (__int128)t[0] * f.v[0] + (__int128)t[1] * g.v[0];
    __int128 cg = (__int128)t[2] * f.v[0] + (__int128)t[3] * g.v[0];
    cf >>= 62; cg >>= 62;                                                       // the low 62 bits are zero
#pragma unroll
    for (int i = 1; i < 7; i++) {
        cf += (__int128)t[0] * f.v[i] + (__int128)t[1] * g.v[i];
        cg += (__int128)t[2] * f.v[i] + (__int128)t[3] * g.v[i];
        f.v[i - 1] = (int64_t)cf & M62; cf >>= 62;
        g.v[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f.v[6] = (int64_t)cf;
    g.v[6] = (int64_t)cg;
}
BLS_HD void inv62_update_de(inv62& d, inv62& e, const int64_t (&t)[4], const inv62& pm, uint64_t pinv62) {
    const int64_t M62 = (int64_t)((1ull << 62) - 1);
    const int64_t sd = d.v[6] >> 63, se = e.v[6] >> 63;
    int64_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);      // start from the multiple of p that keeps d, e in (-2p, p)
    __int128 cd = (__int128)t[0] * d.v[0] + (__int128)t[1] * e.v[0];
    __int128 ce = (__int128)t[2] * d.v[0] + (__int128)t[3] * e.v[0];
    md -= (int64_t)((pinv62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);     // t (d, e) + p (md, me) = 0 mod 2^62
    me -= (int64_t)((pinv62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    cd += (__int128)pm.v[0] * md;
    ce += (__int128)pm.v[0] * me;
    cd >>= 62; ce >>= 62;
#pragma unroll
    for (int i = 1; i < 7; i++) {
        cd += (__int128)t[0] * d.v[i] + (__int128)t[1] * e.v[i] + (__int128)pm.v[i] * md;
        ce += (__int128)t[2] * d.v[i] + (__int128)t[3] * e.v[i] + (__int128)pm.v[i] * me;
        d.v[i - 1] = (int64_t)cd & M62; cd >>= 62;
        e.v[i - 1] = (int64_t)ce & M62; ce >>= 62;
    }
    d.v[6] = (int64_t)cd;
    e.v[6] = (int64_t)ce;
}
// 14 canonical 28-bit limbs (value < 2^392) -> 7 limbs of 62 bits
BLS_HD inv62 inv62_from_limbs(const uint32_t (&l)[FP_N]) {
    inv62 r;
#pragma unroll
    for (int j = 0; j < 7; j++) {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < FP_N; i++) {
            const int sh = 28 * i - 62 * j;                                     // position of limb i inside limb j
            if (sh > -28 && sh < 62) v |= sh >= 0 ? (uint64_t)l[i] << sh : (uint64_t)l[i] >> (-sh);
        }
        r.v[j] = (int64_t)(v & ((1ull << 62) - 1));
    }
    return r;
}
BLS_HDN fp fp_inv(const fp& a) {
    fp ac = fp_canon(a);
    uint32_t pl[FP_N];
#pragma unroll
    for (int i = 0; i < FP_N; i++) pl[i] = k::P[i];
    const inv62 pm = inv62_from_limbs(pl);
    uint64_t pinv62 = 1;                                                        // p^-1 mod 2^62 (Newton: the precision doubles)
    for (int i = 0; i < 6; i++) pinv62 *= 2 - (uint64_t)pm.v[0] * pinv62;
    pinv62 &= (1ull << 62) - 1;
    inv62 f = pm, g = inv62_from_limbs(ac.l), d{{0, 0, 0, 0, 0, 0, 0}}, e{{1, 0, 0, 0, 0, 0, 0}};
    int64_t eta = -1;
    for (int it = 0; it < 24; it++) {                                           // (49 * 381 + 57) / 17 = 1102 steps always suffice: 18 batches
        int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) acc |= g.v[i];
        if (acc == 0) break;
        int64_t t[4];
        inv62_divsteps(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], t);
        inv62_update_de(d, e, t, pm, pinv62);
        inv62_update_fg(f, g, t);
    }
    // f = +-1 (or p for a = 0, then d = 0): the inverse is sign(f) d, a value in (-2p, 2p) -> signed 28-bit limbs
    const int64_t sf = f.v[6] >> 63;
    __int128 c = 0;
    uint64_t w[7];
#pragma unroll
    for (int i = 0; i < 6; i++) {                                               // conditional negation, re-carried
        c += (__int128)((d.v[i] ^ sf) - sf);
        w[i] = (uint64_t)c & ((1ull << 62) - 1);
        c >>= 62;
    }
    c += (__int128)((d.v[6] ^ sf) - sf);
    const int64_t top = (int64_t)c;                                             // small and signed (|value| < 2^383)
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N - 1; i++) {
        uint64_t v = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int sh = 62 * j - 28 * i;
            if (sh > -62 && sh < 28) v |= sh >= 0 ? w[j] << sh : w[j] >> (-sh);
        }
        if (62 * 6 - 28 * i < 28) v |= (uint64_t)top << (62 * 6 - 28 * i);
        r.l[i] = (uint32_t)v & FP_MASK;
    }
    r.l[FP_N - 1] = (uint32_t)(int32_t)((int64_t)(w[5] >> (28 * (FP_N - 1) - 62 * 5)) + top * (int64_t)(1ll << (62 * 6 - 28 * (FP_N - 1))));
    BLS_SET_VB(r, 2);
    BLS_SET_LB(r, 0);
    // the argument was a R, so this is 1 / (a R); times R^3 / R (one Montgomery product with R^3 = RR * RR / R) it is R / a
    fp rr = fp_from_const(k::RR);
    return fp_mul(r, fp_mul(rr, rr));
}

// a^((p-3)/4): for a QR this is 1/sqrt(a); for a non-residue (a*t)^2 = -a.
// Device (round 6): ONE hand-allocated assembly statement (tools/gen_pow_asm.py -> build/pow_asm.inc): the exponent is the same for every lane, so the
// 4-bit sliding-window schedule is unrolled into the code and the table of odd powers lives in registers (v60 .. v171) instead of the scratch memory
// fp_pow_sched indexes it in; 172 VGPRs and no AGPRs, so the two-waves-per-SIMD kernels keep their occupancy.  Contract: |a| < 8 p, limbs of at most two
// units (asserted by the host tracker below); result as fp_mul's: canonical limbs, |r| < 2 p.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_POW_NOASM)
#include "../build/pow_asm.inc"
// Expanded in place at every call site (a few per kernel), not an out-of-line function: the statement clobbers v0 .. v171, half of which the
// calling convention makes a callee preserve - as a function it saved them in AGPRs, and in the unified register file of a 256-register kernel
// those start behind the kernel's own allocation (k_hash_map: 288 registers = one wave per SIMD).
__device__ __forceinline__ fp fp_recip_sqrt_pow(const fp& a) {
    uint32_t a0 = a.l[0], a1 = a.l[1], a2 = a.l[2], a3 = a.l[3], a4 = a.l[4], a5 = a.l[5], a6 = a.l[6], a7 = a.l[7], a8 = a.l[8], a9 = a.l[9], a10 = a.l[10],
             a11 = a.l[11], a12 = a.l[12], a13 = a.l[13];
    asm volatile(BLS_POW_ASM_BODY
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13)
                 :
                 : BLS_POW_ASM_CLOBBERS);
    return fp{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}};
}
#else
BLS_HD fp fp_recip_sqrt_pow(const fp& a) {
    BLS_REQUIRE(BLS_VB(a) <= 8 && BLS_LB(a) <= 2, "fp_recip_sqrt_pow: the device body's input contract");
    return fp_pow_sched(a, k::SW_PM3D4, k::SW_PM3D4_LEN);
}
#endif

// 48 little-endian bytes (blst_fp memory image) <-> fp   (host-side tests and byte-addressed inputs)
BLS_HD fp fp_load_le(const uint8_t* p) {
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 12; i++)
        w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    return fp_from_blst(w);
}

BLS_HD void fp_store_le(uint8_t* p, const fp& a) {
    uint32_t w[12];
    fp_to_blst(w, a);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p[4 * i] = (uint8_t)w[i];
        p[4 * i + 1] = (uint8_t)(w[i] >> 8);
        p[4 * i + 2] = (uint8_t)(w[i] >> 16);
        p[4 * i + 3] = (uint8_t)(w[i] >> 24);
    }
}

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2+1); memory order (c0 real, c1 imaginary) = blst_fp2 (blst_abi.nim:96-98)
// ---------------------------------------------------------------------------------------------
struct fp2 {
    fp c0, c1;
};

BLS_HD fp2 fp2_from_const(const uint32_t (&c)[2 * FP_N]) {
    fp2 r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        r.c0.l[i] = c[i];
        r.c1.l[i] = c[FP_N + i];
    }
    BLS_SET_VB(r.c0, 1);
    BLS_SET_VB(r.c1, 1);
    BLS_SET_LB(r.c0, 1);
    BLS_SET_LB(r.c1, 1);
    return r;
}

BLS_HD fp2 fp2_zero() { return fp2{fp_zero(), fp_zero()}; }
BLS_HD fp2 fp2_one() { return fp2{fp_one(), fp_zero()}; }
BLS_HD bool fp2_is_zero(const fp2& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
BLS_HD bool fp2_eq(const fp2& a, const fp2& b) { return fp_eq(a.c0, b.c0) & fp_eq(a.c1, b.c1); }
BLS_HD fp2 fp2_select(bool c, const fp2& a, const fp2& b) { return fp2{fp_select(c, a.c0, b.c0), fp_select(c, a.c1, b.c1)}; }
BLS_HD fp2 fp2_add(const fp2& a, const fp2& b) { return fp2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
BLS_HD fp2 fp2_sub(const fp2& a, const fp2& b) { return fp2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
BLS_HD fp2 fp2_neg(const fp2& a) { return fp2{fp_neg(a.c0), fp_neg(a.c1)}; }
BLS_HD fp2 fp2_dbl(const fp2& a) { return fp2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
BLS_HD fp2 fp2_add_nc(const fp2& a, const fp2& b) { return fp2{fp_add_nc(a.c0, b.c0), fp_add_nc(a.c1, b.c1)}; }
BLS_HD fp2 fp2_sub_nc(const fp2& a, const fp2& b) { return fp2{fp_sub_nc(a.c0, b.c0), fp_sub_nc(a.c1, b.c1)}; }
BLS_HD fp2 fp2_dbl_nc(const fp2& a) { return fp2{fp_dbl_nc(a.c0), fp_dbl_nc(a.c1)}; }
BLS_HD fp2 fp2_carry(const fp2& a) { return fp2{fp_carry(a.c0), fp_carry(a.c1)}; }
BLS_HD fp2 fp2_conj(const fp2& a) { return fp2{a.c0, fp_neg(a.c1)}; }
BLS_HD fp2 fp2_mul3(const fp2& a) { return fp2_add(fp2_dbl(a), a); }
BLS_HD fp2 fp2_reduce(const fp2& a) { return fp2{fp_reduce(a.c0), fp_reduce(a.c1)}; }


#if defined(__HIP_DEVICE_COMPILE__)
// Whole Fp2 operations as ONE out-of-line call each (results return in v0..v27 as a 32-wide vector; a 28-word
// struct would go through memory).
typedef uint32_t bls_u32x32 __attribute__((ext_vector_type(32)));
__device__ __forceinline__ bls_u32x32 fp2_pack(const fp& c0, const fp& c1) {
    bls_u32x32 r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        r[i] = c0.l[i];
        r[FP_N + i] = c1.l[i];
    }
    r[28] = 0; r[29] = 0; r[30] = 0; r[31] = 0;
    return r;
}
// Second operand of fp2_mul_regs: 28 words per lane handed over through LDS (7 x 16 bytes, [group][lane] so a
// wave's accesses are conflict-free).  Registers carry only 32 argument words; the rest would travel on the
// stack, i.e. through scratch memory, which at 1024 waves x 13 products per line was ~6 GB of HBM writes per
// k_lineprod launch.  Every kernel of this library runs one wave per workgroup, so the slot is wave-private.
// bls_xchg is the hand-over slot of fp2_mul; a kernel may park operands it uses repeatedly in LDS slots of its own
// (k_lineprod: the line coefficients, which then occupy no registers at all) and multiply by them with
// fp2_mul_lds.  A slot is 7 groups of 64 lanes x 16 bytes.
typedef uint32_t bls_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bls_u32x4 bls_lds_u32x4;
constexpr int BLS_LDS_SLOT = 7 * 64;
static __shared__ bls_u32x4 bls_xchg[BLS_LDS_SLOT];
__device__ __noinline__ bls_u32x32 fp2_mul_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15, uint32_t a16, uint32_t a17, uint32_t a18, uint32_t a19, uint32_t a20, uint32_t a21, uint32_t a22, uint32_t a23, uint32_t a24, uint32_t a25, uint32_t a26, uint32_t a27, const bls_lds_u32x4* slot) {
    fp x0{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}}, x1{{a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27}}, y0, y1, nx1;
    uint32_t yw[28];
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        bls_u32x4 v = slot[q * 64 + lane];
        yw[4 * q] = v.x; yw[4 * q + 1] = v.y; yw[4 * q + 2] = v.z; yw[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        y0.l[i] = yw[i];
        y1.l[i] = yw[FP_N + i];
        nx1.l[i] = 0u - x1.l[i];
    }
    return fp2_pack(fp_dot2_core(x0, y0, nx1, y1), fp_dot2_core(x0, y1, x1, y0));
}
__device__ __noinline__ bls_u32x32 fp2_sqr_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15, uint32_t a16, uint32_t a17, uint32_t a18, uint32_t a19, uint32_t a20, uint32_t a21, uint32_t a22, uint32_t a23, uint32_t a24, uint32_t a25, uint32_t a26, uint32_t a27) {
    fp x0{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}}, x1{{a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27}}, sm, df, d0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        sm.l[i] = x0.l[i] + x1.l[i];
        df.l[i] = x0.l[i] - x1.l[i];
        d0.l[i] = 2 * x0.l[i];
    }
    return fp2_pack(fp_mul_core(sm, df), fp_mul_core(d0, x1));
}
// ONE lazily reduced dot product x0 y0 + x1 y1 (half an Fp2 product), (y0, y1) handed over through the LDS slot like the
// second operand of fp2_mul_regs: what a lane of a cooperative team computes when an Fp2 product is split over two lanes.
__device__ __noinline__ fp fp_dot2_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t a12, uint32_t a13, uint32_t a14, uint32_t a15, uint32_t a16, uint32_t a17, uint32_t a18, uint32_t a19, uint32_t a20, uint32_t a21, uint32_t a22, uint32_t a23, uint32_t a24, uint32_t a25, uint32_t a26, uint32_t a27, const bls_lds_u32x4* slot) {
    fp x0{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13}}, x1{{a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27}}, y0, y1;
    uint32_t yw[28];
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        bls_u32x4 v = slot[q * 64 + lane];
        yw[4 * q] = v.x; yw[4 * q + 1] = v.y; yw[4 * q + 2] = v.z; yw[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        y0.l[i] = yw[i];
        y1.l[i] = yw[FP_N + i];
    }
    return fp_dot2_core(x0, y0, x1, y1);
}
#endif

// Fp2 product: c0 = a0 b0 - a1 b1 and c1 = a0 b1 + a1 b0 as two lazily reduced dot products (4 operand products,
// 2 reductions: the multiply-add count of Karatsuba's 3 full products without its additions and carries).
// Operands: at most 2 limb units each (limb-wise sums of two carried values are fine).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ fp2 fp2_unpack(const bls_u32x32& r) {
    fp2 o;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        o.c0.l[i] = r[i];
        o.c1.l[i] = r[FP_N + i];
    }
    return o;
}
__device__ __forceinline__ void fp2_lds_put(bls_lds_u32x4* slot, const fp2& b) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t yw[28];
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        yw[i] = b.c0.l[i];
        yw[FP_N + i] = b.c1.l[i];
    }
#pragma unroll
    for (int q = 0; q < 7; q++) {
        bls_u32x4 v = {yw[4 * q], yw[4 * q + 1], yw[4 * q + 2], yw[4 * q + 3]};
        slot[q * 64 + lane] = v;
    }
}
__device__ __forceinline__ fp2 fp2_lds_get(const bls_lds_u32x4* slot) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t yw[28];
#pragma unroll
    for (int q = 0; q < 7; q++) {
        bls_u32x4 v = slot[q * 64 + lane];
        yw[4 * q] = v.x; yw[4 * q + 1] = v.y; yw[4 * q + 2] = v.z; yw[4 * q + 3] = v.w;
    }
    fp2 r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        r.c0.l[i] = yw[i];
        r.c1.l[i] = yw[FP_N + i];
    }
    return r;
}
// a * (the Fp2 value parked in the LDS slot)
__device__ __forceinline__ fp2 fp2_mul_lds(const fp2& a, const bls_lds_u32x4* slot) {
    return fp2_unpack(fp2_mul_regs(a.c0.l[0], a.c0.l[1], a.c0.l[2], a.c0.l[3], a.c0.l[4], a.c0.l[5], a.c0.l[6], a.c0.l[7], a.c0.l[8], a.c0.l[9], a.c0.l[10], a.c0.l[11], a.c0.l[12], a.c0.l[13], a.c1.l[0], a.c1.l[1], a.c1.l[2], a.c1.l[3], a.c1.l[4], a.c1.l[5], a.c1.l[6], a.c1.l[7], a.c1.l[8], a.c1.l[9], a.c1.l[10], a.c1.l[11], a.c1.l[12], a.c1.l[13], slot));
}
// x0 y0 + x1 y1
__device__ __forceinline__ fp fp_dot2(const fp& x0, const fp& y0, const fp& x1, const fp& y1) {
    bls_lds_u32x4* x = (bls_lds_u32x4*)bls_xchg;
    fp2_lds_put(x, fp2{y0, y1});
    return fp_dot2_regs(x0.l[0], x0.l[1], x0.l[2], x0.l[3], x0.l[4], x0.l[5], x0.l[6], x0.l[7], x0.l[8], x0.l[9], x0.l[10], x0.l[11], x0.l[12], x0.l[13], x1.l[0], x1.l[1], x1.l[2], x1.l[3], x1.l[4], x1.l[5], x1.l[6], x1.l[7], x1.l[8], x1.l[9], x1.l[10], x1.l[11], x1.l[12], x1.l[13], x);
}
__device__ __forceinline__ fp2 fp2_mul(const fp2& a, const fp2& b) {
    bls_lds_u32x4* x = (bls_lds_u32x4*)bls_xchg;
    fp2_lds_put(x, b);
    return fp2_mul_lds(a, x);
}
__device__ __forceinline__ fp2 fp2_sqr(const fp2& a) {
    return fp2_unpack(fp2_sqr_regs(a.c0.l[0], a.c0.l[1], a.c0.l[2], a.c0.l[3], a.c0.l[4], a.c0.l[5], a.c0.l[6], a.c0.l[7], a.c0.l[8], a.c0.l[9], a.c0.l[10], a.c0.l[11], a.c0.l[12], a.c0.l[13], a.c1.l[0], a.c1.l[1], a.c1.l[2], a.c1.l[3], a.c1.l[4], a.c1.l[5], a.c1.l[6], a.c1.l[7], a.c1.l[8], a.c1.l[9], a.c1.l[10], a.c1.l[11], a.c1.l[12], a.c1.l[13]));
}
#else
BLS_HD fp2 fp2_mul(const fp2& a, const fp2& b) {
    return fp2{fp_dot2(a.c0, b.c0, fp_neg(a.c1), b.c1), fp_dot2(a.c0, b.c1, a.c1, b.c0)};
}
// complex squaring: 2 base multiplications, operand sums without carries (inputs: carried values)
BLS_HD fp2 fp2_sqr(const fp2& a) {
    return fp2{fp_mul(fp_add_nc(a.c0, a.c1), fp_sub_nc(a.c0, a.c1)), fp_mul(fp_dbl_nc(a.c0), a.c1)};
}
#endif

// The same two operations with the multiplier bodies expanded IN PLACE (device: no call, no argument moves, no LDS hand-over of the
// second operand; ~1 350 / ~960 instructions of code per use): for the few loops whose whole body may be this large - the doubling
// runs of the cofactor clearing, where an out-of-line call costs 56 register moves per product and keeps the loop-carried point out
// of the multiplier's registers.  Host: the checked out-of-line forms (same arithmetic, same census).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ fp2 fp2_mul_inl(const fp2& a, const fp2& b) {
    return fp2{fp_dot2_core(a.c0, b.c0, fp_neg(a.c1), b.c1), fp_dot2_core(a.c0, b.c1, a.c1, b.c0)};
}
__device__ __forceinline__ fp2 fp2_sqr_inl(const fp2& a) {
    return fp2{fp_mul_core(fp_add_nc(a.c0, a.c1), fp_sub_nc(a.c0, a.c1)), fp_mul_core(fp_dbl_nc(a.c0), a.c1)};
}
__device__ __forceinline__ fp fp_mul_inl(const fp& a, const fp& b) { return fp_mul_core(a, b); }
__device__ __forceinline__ fp fp_sqr_inl(const fp& a) { return fp_sqr_core(a); }
#else
BLS_HD fp2 fp2_mul_inl(const fp2& a, const fp2& b) { return fp2_mul(a, b); }
BLS_HD fp2 fp2_sqr_inl(const fp2& a) { return fp2_sqr(a); }
BLS_HD fp fp_mul_inl(const fp& a, const fp& b) { return fp_mul(a, b); }
BLS_HD fp fp_sqr_inl(const fp& a) { return fp_sqr(a); }
#endif

BLS_HD fp2 fp2_mul_fp(const fp2& a, const fp& b) { return fp2{fp_mul(a.c0, b), fp_mul(a.c1, b)}; }

// multiply by the sextic non-residue xi = 1+u
BLS_HD fp2 fp2_mul_xi(const fp2& a) { return fp2{fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }
BLS_HD fp2 fp2_mul_xi_nc(const fp2& a) { return fp2{fp_sub_nc(a.c0, a.c1), fp_add_nc(a.c0, a.c1)}; }

BLS_HD fp fp2_norm(const fp2& a) { return fp_add(fp_sqr(a.c0), fp_sqr(a.c1)); }

BLS_HD fp2 fp2_inv(const fp2& a) {
    fp n = fp_inv(fp2_norm(a));
    return fp2{fp_mul(a.c0, n), fp_neg(fp_mul(a.c1, n))};
}

// RFC 9380 section 4.1 sgn0 for m = 2 (on canonical, non-Montgomery values)
BLS_HD uint32_t fp2_sgn0(const fp2& a) {
    fp x0 = fp_from_mont(a.c0);
    fp x1 = fp_from_mont(a.c1);
    uint32_t s0 = x0.l[0] & 1, s1 = x1.l[0] & 1;
    uint32_t z0 = fp_limbs_are_zero(x0) ? 1u : 0u;
    return s0 | (z0 & s1);
}

BLS_HD fp2 fp2_load_le(const uint8_t* p) { return fp2{fp_load_le(p), fp_load_le(p + 48)}; }
BLS_HD void fp2_store_le(uint8_t* p, const fp2& a) {
    fp_store_le(p, a.c0);
    fp_store_le(p + 48, a.c1);
}

}  // namespace bls
