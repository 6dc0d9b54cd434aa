// Fp = GF(p), BLS12-381 base field, for gfx950 lanes.
//
// Representation: 12 x 32-bit little-endian limbs in Montgomery form, R = 2^384 -- byte-identical
// to the reference's in-memory blst_fp (6 x u64 LE Montgomery limbs, blst_abi.nim:87-94), so
// SignatureSet records are consumed without conversion.  Values are kept fully reduced (< p).
// The inner product step is a 32x32+64 multiply-add (v_mad_u64_u32 on CDNA4); MFMA is not used:
// carry-propagated multi-precision arithmetic is not a dense contraction.
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

namespace bls {

struct fp {
    uint32_t l[12];
};

BLS_HD fp fp_from_const(const uint32_t (&c)[12]) {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = c[i];
    return r;
}

BLS_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = 0;
    return r;
}

BLS_HD fp fp_one() { return fp_from_const(k::ONE); }

BLS_HD bool fp_is_zero(const fp& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc |= a.l[i];
    return acc == 0;
}

BLS_HD bool fp_eq(const fp& a, const fp& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc |= a.l[i] ^ b.l[i];
    return acc == 0;
}

// r = c ? a : b
BLS_HD fp fp_select(bool c, const fp& a, const fp& b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

// t (12 limbs + carry bit) -> t mod p, given t < 2p
BLS_HD fp fp_reduce_once(const uint32_t (&t)[12], uint32_t top) {
    uint32_t d[12];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t v = (uint64_t)t[i] - k::P[i] - borrow;
        d[i] = (uint32_t)v;
        borrow = (v >> 32) & 1;
    }
    // t >= p  <=>  top set, or no final borrow
    bool ge = top != 0 || borrow == 0;
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = ge ? d[i] : t[i];
    return r;
}

BLS_HD fp fp_add(const fp& a, const fp& b) {
    uint32_t t[12];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t v = (uint64_t)a.l[i] + b.l[i] + c;
        t[i] = (uint32_t)v;
        c = v >> 32;
    }
    return fp_reduce_once(t, (uint32_t)c);
}

BLS_HD fp fp_sub(const fp& a, const fp& b) {
    uint32_t t[12];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t v = (uint64_t)a.l[i] - b.l[i] - borrow;
        t[i] = (uint32_t)v;
        borrow = (v >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    fp r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t v = (uint64_t)t[i] + (k::P[i] & mask) + c;
        r.l[i] = (uint32_t)v;
        c = v >> 32;
    }
    return r;
}

BLS_HD fp fp_neg(const fp& a) {
    fp z = fp_zero();
    return fp_sub(z, a);
}

BLS_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }

#if defined(__HIP_DEVICE_COMPILE__)
#include "fp_mul_gfx950.inc"
// The multiplier body (~770 instructions, ~5.5 KB) is shared by every caller so that hot loops fit the
// 64 KB instruction cache.  Operands travel in VGPRs: 24 scalar parameters map to v0..v23 and the
// result returns in v0..v11 (aggregate by-reference parameters would go through scratch memory,
// which at 65 536 lanes no longer fits L2).
__device__ __noinline__ fp fp_mul_regs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7,
                                       uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3,
                                       uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
    fp a{{a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11}}, b{{b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11}};
    return fp_mul_gfx950(a, b);
}
#endif

// Montgomery product a*b*R^-1 mod p.  Device: product-scanning columns of v_mad_u64_u32 +
// v_addc_co_u32 pairs (fp_mul_gfx950.inc).  Host (tests/host_emu only): portable CIOS with 32-bit
// limbs; top word of p < 2^31 so 13 words suffice.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ fp fp_mul(const fp& a, const fp& b) {
    return fp_mul_regs(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11],
                       b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
}
#else
__host__ __noinline__ inline fp fp_mul(const fp& a, const fp& b) {
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint32_t bi = b.l[i];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint64_t v = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)v;
            c = v >> 32;
        }
        uint64_t v = (uint64_t)t[12] + c;
        t[12] = (uint32_t)v;
        const uint32_t m = t[0] * k::N0;
        c = ((uint64_t)m * k::P[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 12; j++) {
            uint64_t w = (uint64_t)m * k::P[j] + t[j] + c;
            t[j - 1] = (uint32_t)w;
            c = w >> 32;
        }
        uint64_t w = (uint64_t)t[12] + c;
        t[11] = (uint32_t)w;
        t[12] = (uint32_t)(w >> 32);
    }
    uint32_t lo[12];
#pragma unroll
    for (int i = 0; i < 12; i++) lo[i] = t[i];
    return fp_reduce_once(lo, t[12]);
}
#endif

BLS_HD fp fp_sqr(const fp& a) { return fp_mul(a, a); }

// a * small constant via additions
BLS_HD fp fp_mul3(const fp& a) { return fp_add(fp_dbl(a), a); }

// Montgomery -> canonical integer limbs (multiply by 1)
BLS_HD fp fp_from_mont(const fp& a) {
    fp one = fp_zero();
    one.l[0] = 1;
    return fp_mul(a, one);
}

BLS_HD fp fp_to_mont(const fp& a) { return fp_mul(a, fp_from_const(k::RR)); }

// a^e for a 384-bit exponent given as 12 LE limbs; fixed 4-bit window, not constant time
// (nothing secret on this path: public keys, messages, signatures and public blinding scalars).
BLS_HDN fp fp_pow(const fp& a, const uint32_t (&e)[12]) {
    fp tab[16];
    tab[0] = fp_one();
    tab[1] = a;
    for (int i = 2; i < 16; i++) tab[i] = fp_mul(tab[i - 1], a);
    fp r = fp_one();
    bool started = false;
    for (int w = 95; w >= 0; w--) {
        uint32_t nib = (e[w >> 3] >> ((w & 7) * 4)) & 0xf;
        if (started) {
            r = fp_sqr(r);
            r = fp_sqr(r);
            r = fp_sqr(r);
            r = fp_sqr(r);
        }
        if (nib) {
            r = started ? fp_mul(r, tab[nib]) : tab[nib];
            started = true;
        }
    }
    return r;
}

BLS_HD fp fp_inv(const fp& a) {
    const uint32_t e[12] = {k::EXP_PM2[0], k::EXP_PM2[1], k::EXP_PM2[2], k::EXP_PM2[3], k::EXP_PM2[4], k::EXP_PM2[5],
                            k::EXP_PM2[6], k::EXP_PM2[7], k::EXP_PM2[8], k::EXP_PM2[9], k::EXP_PM2[10], k::EXP_PM2[11]};
    return fp_pow(a, e);
}

// a^((p-3)/4): for a QR this is 1/sqrt(a); for a non-residue (a*t)^2 = -a.
BLS_HD fp fp_recip_sqrt_pow(const fp& a) {
    const uint32_t e[12] = {k::EXP_PM3D4[0], k::EXP_PM3D4[1], k::EXP_PM3D4[2], k::EXP_PM3D4[3], k::EXP_PM3D4[4], k::EXP_PM3D4[5],
                            k::EXP_PM3D4[6], k::EXP_PM3D4[7], k::EXP_PM3D4[8], k::EXP_PM3D4[9], k::EXP_PM3D4[10], k::EXP_PM3D4[11]};
    return fp_pow(a, e);
}

// 48 little-endian bytes (blst_fp memory image) <-> fp
BLS_HD fp fp_load_le(const uint8_t* p) {
    fp r;
#pragma unroll
    for (int i = 0; i < 12; i++)
        r.l[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    return r;
}

BLS_HD void fp_store_le(uint8_t* p, const fp& a) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p[4 * i] = (uint8_t)a.l[i];
        p[4 * i + 1] = (uint8_t)(a.l[i] >> 8);
        p[4 * i + 2] = (uint8_t)(a.l[i] >> 16);
        p[4 * i + 3] = (uint8_t)(a.l[i] >> 24);
    }
}

// ---------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2+1); memory order (c0 real, c1 imaginary) = blst_fp2 (blst_abi.nim:96-98)
// ---------------------------------------------------------------------------------------------
struct fp2 {
    fp c0, c1;
};

BLS_HD fp2 fp2_from_const(const uint32_t (&c)[24]) {
    fp2 r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        r.c0.l[i] = c[i];
        r.c1.l[i] = c[12 + i];
    }
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
BLS_HD fp2 fp2_conj(const fp2& a) { return fp2{a.c0, fp_neg(a.c1)}; }
BLS_HD fp2 fp2_mul3(const fp2& a) { return fp2_add(fp2_dbl(a), a); }

// Karatsuba: 3 base multiplications
BLS_HD fp2 fp2_mul(const fp2& a, const fp2& b) {
    fp t0 = fp_mul(a.c0, b.c0);
    fp t1 = fp_mul(a.c1, b.c1);
    fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return fp2{fp_sub(t0, t1), fp_sub(fp_sub(s, t0), t1)};
}

// complex squaring: 2 base multiplications
BLS_HD fp2 fp2_sqr(const fp2& a) {
    fp t = fp_mul(a.c0, a.c1);
    fp c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
    return fp2{c0, fp_dbl(t)};
}

BLS_HD fp2 fp2_mul_fp(const fp2& a, const fp& b) { return fp2{fp_mul(a.c0, b), fp_mul(a.c1, b)}; }

// multiply by the sextic non-residue xi = 1+u
BLS_HD fp2 fp2_mul_xi(const fp2& a) { return fp2{fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; }

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
    uint32_t z0 = fp_is_zero(x0) ? 1u : 0u;
    return s0 | (z0 & s1);
}

BLS_HD fp2 fp2_load_le(const uint8_t* p) { return fp2{fp_load_le(p), fp_load_le(p + 48)}; }
BLS_HD void fp2_store_le(uint8_t* p, const fp2& a) {
    fp_store_le(p, a.c0);
    fp_store_le(p + 48, a.c1);
}

}  // namespace bls
