// HIP kernels + C ABI of the MI355X batch-verification path (gfx950 only).
//
// Pipeline for one batch of n SignatureSets (reference call stack: bls_batch_verifier.nim:296-371 ->
// blst_min_pubkey_sig_core.nim:476-568,649-672 -> BLST):
//   k_blind       one lane per blinding chain ("virtual thread"): r_i                  (core :497-507,:545-556)
//   k_hash_map    two lanes per tuple: hash_to_field, SSWU + 3-isogeny of u_0 / u_1      (blst hash part)
//   k_hash_clear  one lane per tuple: sum of the two mapped points, cofactor clearing; Jacobian H_i
//   k_pkmul       one lane per tuple: [r_i]PK_i (signed 4-bit windows), Jacobian; infinity-pk flag (blst pk part)
//   signature side (blst sig part + finalverify's extra pair):
//     k_sig_convert, k_msm_hist/scan/scatter (counting sort by digit of r_i), k_sig_bucket: bucket
//     sums B_{w,d} -> extra Miller pairs (-[d 2^(cw)]G1, B_{w,d})
//   k_lines       one lane per pair: 68 Miller lines -> HBM, step-major SoA             (miller_loop_n)
//   k_lineprod    (step, pair-range) grid: per-lane sparse products, wave-shuffle Fp12 product tree
//   k_lineprod2   per step: product of the range partials -> L_s
//   k_tail        one wave, lane-parallel Fp12: Horner over the 68 L_s, conjugate, [shard merge],
//                 final exponentiation, == 1
// Intermediates live in HBM as structure-of-arrays of 16-byte limb groups so that lane i's
// loads/stores of one limb group are contiguous across the wave (coalesced dwordx4).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/blscurve_mi355x.h"
#include "deser.hpp"
#include "h2c.hpp"
#include "pairing.hpp"
#include "c12.hpp"

using namespace bls;

namespace {

thread_local std::string g_err;

#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                \
            return MI355_BLS_ERR_HIP;                                                              \
        }                                                                                          \
    } while (0)

constexpr int WAVE = 64;

struct dst_t {
    uint8_t b[64];
    uint32_t len;
};

// ------------------------------------------------------------------------------------------
// SoA accessors: an Fp element is 14 limbs = 4 uint4 (2 pad words); plane p of element i lives at
// base[(4p+q)*stride + i], q = 0..3
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ fp soa_ld(const uint4* base, size_t stride, uint32_t plane, size_t i) {
    fp r;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint4 v = base[(size_t)(plane * 4 + q) * stride + i];
        r.l[4 * q] = v.x;
        r.l[4 * q + 1] = v.y;
        if (q < 3) {
            r.l[4 * q + 2] = v.z;
            r.l[4 * q + 3] = v.w;
        }
    }
    return r;
}
__device__ __forceinline__ void soa_st(uint4* base, size_t stride, uint32_t plane, size_t i, const fp& a) {
#pragma unroll
    for (int q = 0; q < 4; q++)
        base[(size_t)(plane * 4 + q) * stride + i] =
            q < 3 ? make_uint4(a.l[4 * q], a.l[4 * q + 1], a.l[4 * q + 2], a.l[4 * q + 3]) : make_uint4(a.l[12], a.l[13], 0u, 0u);
}
__device__ __forceinline__ fp2 soa_ld2(const uint4* base, size_t stride, uint32_t plane, size_t i) {
    return fp2{soa_ld(base, stride, plane, i), soa_ld(base, stride, plane + 1, i)};
}
__device__ __forceinline__ void soa_st2(uint4* base, size_t stride, uint32_t plane, size_t i, const fp2& a) {
    soa_st(base, stride, plane, i, a.c0);
    soa_st(base, stride, plane + 1, i, a.c1);
}
__device__ __forceinline__ g2_jac soa_ld_g2(const uint4* base, size_t stride, size_t i) {
    return g2_jac{soa_ld2(base, stride, 0, i), soa_ld2(base, stride, 2, i), soa_ld2(base, stride, 4, i)};
}
__device__ __forceinline__ void soa_st_g2(uint4* base, size_t stride, size_t i, const g2_jac& a) {
    soa_st2(base, stride, 0, i, a.x);
    soa_st2(base, stride, 2, i, a.y);
    soa_st2(base, stride, 4, i, a.z);
}
__device__ __forceinline__ g1_jac soa_ld_g1(const uint4* base, size_t stride, size_t i) {
    return g1_jac{soa_ld(base, stride, 0, i), soa_ld(base, stride, 1, i), soa_ld(base, stride, 2, i)};
}
__device__ __forceinline__ void soa_st_g1(uint4* base, size_t stride, size_t i, const g1_jac& a) {
    soa_st(base, stride, 0, i, a.x);
    soa_st(base, stride, 1, i, a.y);
    soa_st(base, stride, 2, i, a.z);
}

// Internal AoS buffers (partials, step products): FPW words per Fp (14 limbs + 2 pad), device representation.
constexpr int FPW = 16, G1W = 3 * FPW, G2W = 6 * FPW, F12W = 12 * FPW;
__device__ __forceinline__ fp ld_fp_int(const uint32_t* w) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = w[i];
    return r;
}
__device__ __forceinline__ void st_fp_int(uint32_t* w, const fp& a) {
#pragma unroll
    for (int i = 0; i < FP_N; i++) w[i] = a.l[i];
}
__device__ __forceinline__ g1_jac ld_g1_int(const uint32_t* w) { return g1_jac{ld_fp_int(w), ld_fp_int(w + FPW), ld_fp_int(w + 2 * FPW)}; }
__device__ __forceinline__ void st_g1_int(uint32_t* w, const g1_jac& a) {
    st_fp_int(w, a.x); st_fp_int(w + FPW, a.y); st_fp_int(w + 2 * FPW, a.z);
}
__device__ __forceinline__ g2_jac ld_g2_int(const uint32_t* w) {
    return g2_jac{fp2{ld_fp_int(w), ld_fp_int(w + FPW)}, fp2{ld_fp_int(w + 2 * FPW), ld_fp_int(w + 3 * FPW)},
                  fp2{ld_fp_int(w + 4 * FPW), ld_fp_int(w + 5 * FPW)}};
}
__device__ __forceinline__ void st_g2_int(uint32_t* w, const g2_jac& a) {
    st_fp_int(w, a.x.c0); st_fp_int(w + FPW, a.x.c1); st_fp_int(w + 2 * FPW, a.y.c0);
    st_fp_int(w + 3 * FPW, a.y.c1); st_fp_int(w + 4 * FPW, a.z.c0); st_fp_int(w + 5 * FPW, a.z.c1);
}
__device__ __forceinline__ void st_fp12_int(uint32_t* w, const fp12& a) {
    const fp2* c[6] = {&a.c0.a0, &a.c0.a1, &a.c0.a2, &a.c1.a0, &a.c1.a1, &a.c1.a2};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        st_fp_int(w + 2 * FPW * i, c[i]->c0);
        st_fp_int(w + 2 * FPW * i + FPW, c[i]->c1);
    }
}
__device__ __forceinline__ fp12 ld_fp12_int(const uint32_t* w) {
    fp12 a;
    fp2* c[6] = {&a.c0.a0, &a.c0.a1, &a.c0.a2, &a.c1.a0, &a.c1.a1, &a.c1.a2};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        c[i]->c0 = ld_fp_int(w + 2 * FPW * i);
        c[i]->c1 = ld_fp_int(w + 2 * FPW * i + FPW);
    }
    return a;
}

// The reference's memory images (blst_fp: 12 words, Montgomery R = 2^384; u64-limb structs, 8-byte aligned):
// converted to / from the device representation with one multiplication per element.
__device__ __forceinline__ fp ld_fp_blst(const uint32_t* w) {
    uint32_t t[12];
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = w[i];
    return fp_from_blst(t);
}
__device__ __forceinline__ void st_fp_blst(uint32_t* w, const fp& a) {
    uint32_t t[12];
    fp_to_blst(t, a);
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = t[i];
}
__device__ __forceinline__ g1_aff ld_g1a_blst(const uint32_t* w) { return g1_aff{ld_fp_blst(w), ld_fp_blst(w + 12)}; }
__device__ __forceinline__ g2_aff ld_g2a_blst(const uint32_t* w) {
    return g2_aff{fp2{ld_fp_blst(w), ld_fp_blst(w + 12)}, fp2{ld_fp_blst(w + 24), ld_fp_blst(w + 36)}};
}
__device__ __forceinline__ g1_jac ld_g1_blst(const uint32_t* w) { return g1_jac{ld_fp_blst(w), ld_fp_blst(w + 12), ld_fp_blst(w + 24)}; }
__device__ __forceinline__ g2_jac ld_g2_blst(const uint32_t* w) {
    return g2_jac{fp2{ld_fp_blst(w), ld_fp_blst(w + 12)}, fp2{ld_fp_blst(w + 24), ld_fp_blst(w + 36)}, fp2{ld_fp_blst(w + 48), ld_fp_blst(w + 60)}};
}
__device__ __forceinline__ void st_g1_blst(uint32_t* w, const g1_jac& a) {
    st_fp_blst(w, a.x); st_fp_blst(w + 12, a.y); st_fp_blst(w + 24, a.z);
}
__device__ __forceinline__ void st_g2_blst(uint32_t* w, const g2_jac& a) {
    st_fp_blst(w, a.x.c0); st_fp_blst(w + 12, a.x.c1); st_fp_blst(w + 24, a.y.c0);
    st_fp_blst(w + 36, a.y.c1); st_fp_blst(w + 48, a.z.c0); st_fp_blst(w + 60, a.z.c1);
}

// wave-level exchange of whole structs through DPP/bpermute shuffles
template <class T>
__device__ __forceinline__ T shfl_down_struct(const T& v, int delta) {
    static_assert(sizeof(T) % 4 == 0, "");
    T r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) d[i] = __shfl_down(s[i], delta, WAVE);
    return r;
}

// broadcast within a group of lanes: value of lane gbase + role
__device__ __forceinline__ fp2 fp2_from_role(const fp2& a, uint32_t gbase, uint32_t role) {
    fp2 r;
    int src = (int)(gbase + role);
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        r.c0.l[i] = __shfl(a.c0.l[i], src, WAVE);
        r.c1.l[i] = __shfl(a.c1.l[i], src, WAVE);
    }
    return r;
}
__device__ __forceinline__ fp fp_from_role(const fp& a, uint32_t gbase, uint32_t role) {
    fp r;
    int src = (int)(gbase + role);
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __shfl(a.l[i], src, WAVE);
    return r;
}

// ------------------------------------------------------------------------------------------
// k_blind: chunk c of B (parallel_chunks.nim:42-66) -> seed = SHA256(rnd || LE64(c)), then per tuple
// seed <- SHA256(seed) until low u64 != 0 (blst_min_pubkey_sig_core.nim:497-507,:545-556).  One lane per chain.
// (batchVerifySerial's single chain is computed on the host: host_serial_chain.)
// Tuples are addressed relative to tuple_base (first tuple of this shard).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) k_blind(const uint8_t* __restrict__ rnd, uint64_t n_total, uint32_t nchunks, uint32_t chunk_lo,
                                                uint32_t chunk_cnt, uint64_t tuple_base, uint64_t tuple_cnt, const uint32_t* __restrict__ carry_in,
                                                uint32_t* __restrict__ carry_out, uint64_t* __restrict__ r_out) {
    // Only the links of tuples [tuple_base, tuple_base + tuple_cnt) are written (a SLICE of the shard: a batch larger than the
    // context's capacity is processed slice by slice).  A chunk that the previous slice cut in two resumes from the chain state
    // that slice left in carry_in (8 seed words); a chunk this slice cuts leaves its state in carry_out (a different buffer: the
    // two may belong to different lanes of this launch).  Slices are processed in order, so at most one chunk is open at a time.
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= chunk_cnt) return;
    uint64_t c = (uint64_t)chunk_lo + t;
    uint64_t off, len;
    uint64_t base = n_total / nchunks, rem = n_total % nchunks;
    if (c < rem) {
        off = (base + 1) * c;
        len = base + 1;
    } else {
        off = base * c + rem;
        len = base;
    }
    const uint64_t t_hi = tuple_base + tuple_cnt;
    uint64_t j0 = off < tuple_base ? tuple_base - off : 0, j1 = off + len > t_hi ? t_hi - off : len;
    uint32_t seed[8];
    if (j0 > 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) seed[i] = carry_in[i];
    } else {
        sha256_ctx ctx;
        sha256_begin(ctx);
        for (int i = 0; i < 32; i++) sha256_put(ctx, rnd[i]);
        for (int i = 0; i < 8; i++) sha256_put(ctx, (uint8_t)(c >> (8 * i)));
        sha256_end(ctx, seed);
    }
    for (uint64_t j = j0; j < j1; j++) {
        uint64_t r;
        do {
            uint32_t nx[8];
            sha256_of_digest(seed, nx);
#pragma unroll
            for (int i = 0; i < 8; i++) seed[i] = nx[i];
            r = digest_low_u64_le(seed);
        } while (r == 0);
        r_out[off + j - tuple_base] = r;
    }
    if (j1 < len) {
#pragma unroll
        for (int i = 0; i < 8; i++) carry_out[i] = seed[i];
    }
}

// Many independent batches in one pass (mi355_bls_batch_verify_many): lane t is chain `c` of batch `b` - batch b has meta[b] =
// {first tuple, tuple count, chains B_b = min(n_b, num_threads), first lane} and its own secureRandomBytes rnds[32 b ..]; the chain
// is exactly the one k_blind computes for that batch alone.  Batches that take the serial chain have B_b = 0 here (host-computed).
struct many_meta {
    uint64_t first, count;
    uint32_t chains, lane0;
};
__global__ void __launch_bounds__(WAVE) k_blind_many(const uint8_t* __restrict__ rnds, const many_meta* __restrict__ meta, uint32_t nbatch, uint32_t nlanes,
                                                     uint64_t* __restrict__ r_out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlanes) return;
    uint32_t lo = 0, hi = nbatch;                        // the batch whose lane range holds t: the last b with lane0 <= t
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (meta[mid].lane0 <= t) lo = mid; else hi = mid;
    }
    const many_meta mb = meta[lo];
    uint64_t c = t - mb.lane0, nchunks = mb.chains, base = mb.count / nchunks, rem = mb.count % nchunks;
    uint64_t off = c < rem ? (base + 1) * c : base * c + rem, len = c < rem ? base + 1 : base;
    const uint8_t* rnd = rnds + 32 * (size_t)lo;
    sha256_ctx ctx;
    sha256_begin(ctx);
    for (int i = 0; i < 32; i++) sha256_put(ctx, rnd[i]);
    for (int i = 0; i < 8; i++) sha256_put(ctx, (uint8_t)(c >> (8 * i)));
    uint32_t seed[8];
    sha256_end(ctx, seed);
    for (uint64_t j = 0; j < len; j++) {
        uint64_t r;
        do {
            uint32_t nx[8];
            sha256_of_digest(seed, nx);
#pragma unroll
            for (int i = 0; i < 8; i++) seed[i] = nx[i];
            r = digest_low_u64_le(seed);
        } while (r == 0);
        r_out[mb.first + off + j] = r;
    }
}

// Batch form of hash-to-G2 in two kernels.  k_hash_map: TWO lanes per message, lane j maps u_j (SSWU + 3-isogeny:
// Fp exponentiations with a small live set), compiled for 256 registers so two waves share a SIMD and fill
// each other's issue gaps; k_hash_clear: one lane per message adds the two points and clears the cofactor
// (G2 arithmetic: needs the full register file).
__global__ void __launch_bounds__(WAVE, 2) k_hash_map(const uint8_t* __restrict__ sets, uint32_t n, dst_t dst, xmd32_consts xc, uint4* __restrict__ M,
                                                         size_t mstride) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 1;
    if (i >= n) return;
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(sets + (size_t)i * 320 + 96);
    fp2 u0, u1;
    if (xc.valid) {                                   // wave-uniform: constants of this DST prepared on the host
        uint32_t mbe[8];
#pragma unroll
        for (int j = 0; j < 8; j++) mbe[j] = bswap32(mw[j]);
        hash_to_field_fp2x2_msg32(u0, u1, mbe, xc);
    } else {
        uint8_t msg[32];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t w = mw[j];
            msg[4 * j] = (uint8_t)w;
            msg[4 * j + 1] = (uint8_t)(w >> 8);
            msg[4 * j + 2] = (uint8_t)(w >> 16);
            msg[4 * j + 3] = (uint8_t)(w >> 24);
        }
        hash_to_field_fp2x2(u0, u1, msg, 32, dst.b, dst.len);
    }
    fp2 u = fp2_select((t & 1) != 0, u1, u0);
    soa_st_g2(M, mstride, t, iso3_g2(sswu_g2(u)));
}
// base point of the doubling chains parked in three LDS slots (21 KB of the 40 KB a wave may use)
#if defined(__HIP_DEVICE_COMPILE__)
struct g2_park_lds {
    bls_lds_u32x4* base;
    __device__ __forceinline__ void put(const g2_jac& a) const {
        fp2_lds_put(base, a.x);
        fp2_lds_put(base + BLS_LDS_SLOT, a.y);
        fp2_lds_put(base + 2 * BLS_LDS_SLOT, a.z);
    }
    __device__ __forceinline__ g2_jac get() const { return g2_jac{fp2_lds_get(base), fp2_lds_get(base + BLS_LDS_SLOT), fp2_lds_get(base + 2 * BLS_LDS_SLOT)}; }
};
#endif
__global__ void __launch_bounds__(WAVE) k_hash_clear(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_jac q0 = soa_ld_g2(M, mstride, 2 * (size_t)i), q1 = soa_ld_g2(M, mstride, 2 * (size_t)i + 1);
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 park_slots[3 * BLS_LDS_SLOT];
    g2_park_lds park{(bls_lds_u32x4*)park_slots};
#else
    g2_park_regs park;               // host pass of the translation unit: kernels are parsed, never run
#endif
    soa_st_g2(H, stride, i, clear_cofactor_g2_with(jac_add(q0, q1), park, BLS_CLEAR_MUL{}));
}

// arbitrary-length message (fastAggregateVerify / coreVerify shape): ONE message, so latency is all that matters.
// A wave works on it cooperatively: the two SSWU maps run in lanes 0 and 1, and every G2 doubling of the cofactor
// clearing (128 of them) spreads the independent products of its first two rounds over lanes 0..2
// (3 multiplication times per doubling instead of 7).  Every lane holds the same points throughout.
// lane-parallel jac_dbl inside a group of 8 lanes that all hold the same point (same formulas, carries and
// reductions as curve.hpp's): roles 0..2 take the three independent products of each of the first two rounds
// device teams: lanes gbase .. gbase + 7 hold the same values; roles 0.. take one product each of a round (ONE multiplier call
// with per-lane operands), then every lane reads all results with wave shuffles.  Formulas: jac_dbl_team / miller_dbl_step_team.
template <int CTRL>
__device__ __forceinline__ fp fp_quad_perm(const fp& a) {
    fp r;
#ifdef BLS_TEAM_SHFL
    const int q = threadIdx.x & 3, src = (threadIdx.x & ~3) | ((CTRL >> (2 * q)) & 3);
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __shfl(a.l[i], src, WAVE);
#else
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], CTRL, 0xf, 0xf, true);
#endif
    return r;
}
// The same interface on SIXTEEN lanes (one DPP row) for the one-message kernel, whose whole wave serves one point: an Fp2 product is FOUR Fp products on four
// lanes (role r: product r / 4, part r % 4 = a0 b0 | a1 b1 | a0 b1 | a1 b0), combined inside the quad with one DPP exchange (real = p0 - p1, imaginary = p2 + p3)
// and partially reduced, so a round costs a 392-multiply-add Fp product instead of the 784 of a lazily reduced half (k_hash_one's mul rounds 4.6 k -> ~3 k cycles).
// Squares are two Fp products already and keep the two-lanes-per-square form.  Results: |value| < 0.51 p, carried limbs - tighter than team_lanes8's.
struct team_lanes16 {
    uint32_t gbase, role;
    __device__ __forceinline__ fp quarter(const fp2& a, const fp2& b) const {
        const uint32_t part = role & 3;
        const bool second = part == 1;
        fp v = fp_mul(fp_select((part & 1) != 0, a.c1, a.c0), fp_select(part == 1 || part == 2, b.c1, b.c0));
        fp w = fp_quad_perm<0xb1>(v);                                      // the partner's product: parts 0 <-> 1, 2 <-> 3
        fp d = fp_sub_nc(fp_select(second, w, v), fp_select(second, v, w));  // p0 - p1 in both lanes of the first pair
        return fp_reduce(fp_select(part >= 2, fp_add_nc(v, w), d));
    }
    __device__ __forceinline__ fp2 gatherq(const fp& v, uint32_t q) const { return fp2{fp_from_role(v, gbase, 4 * q), fp_from_role(v, gbase, 4 * q + 2)}; }
    __device__ __forceinline__ fp2 pick4(const fp2& a0, const fp2& a1, const fp2& a2, const fp2& a3) const {
        const uint32_t q = role >> 2;
        return fp2_select(q < 2, fp2_select(q == 0, a0, a1), fp2_select(q == 2, a2, a3));
    }
    __device__ __forceinline__ void mul4(fp2& r0, fp2& r1, fp2& r2, fp2& r3, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1, const fp2& a2, const fp2& b2,
                                         const fp2& a3, const fp2& b3) const {
        fp v = quarter(pick4(a0, a1, a2, a3), pick4(b0, b1, b2, b3));
        r0 = gatherq(v, 0); r1 = gatherq(v, 1); r2 = gatherq(v, 2); r3 = gatherq(v, 3);
    }
    __device__ __forceinline__ void mul3(fp2& r0, fp2& r1, fp2& r2, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1, const fp2& a2, const fp2& b2) const {
        fp v = quarter(pick4(a0, a1, a2, a2), pick4(b0, b1, b2, b2));
        r0 = gatherq(v, 0); r1 = gatherq(v, 1); r2 = gatherq(v, 2);
    }
    __device__ __forceinline__ void mul2(fp2& r0, fp2& r1, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1) const {
        const bool first = (role >> 2) == 0;
        fp v = quarter(fp2_select(first, a0, a1), fp2_select(first, b0, b1));
        r0 = gatherq(v, 0); r1 = gatherq(v, 1);
    }
    __device__ __forceinline__ fp2 mul1(const fp2& a, const fp2& b) const { return gatherq(quarter(a, b), 0); }
    // squares: role r < 6 takes half r % 2 of square r / 2, as in team_lanes8
    __device__ __forceinline__ fp half_sqr(const fp2& a) const {
        const bool im = (role & 1) != 0;
        return fp_mul(fp_select(im, fp_dbl_nc(a.c0), fp_add_nc(a.c0, a.c1)), fp_select(im, a.c1, fp_sub_nc(a.c0, a.c1)));
    }
    __device__ __forceinline__ fp2 gather2(const fp& v, uint32_t q) const { return fp2{fp_from_role(v, gbase, 2 * q), fp_from_role(v, gbase, 2 * q + 1)}; }
    __device__ __forceinline__ void sqr3(fp2& r0, fp2& r1, fp2& r2, const fp2& a0, const fp2& a1, const fp2& a2) const {
        const uint32_t q = role >> 1;
        fp v = half_sqr(fp2_select(q == 0, a0, fp2_select(q == 1, a1, a2)));
        r0 = gather2(v, 0); r1 = gather2(v, 1); r2 = gather2(v, 2);
    }
    __device__ __forceinline__ void sqr2(fp2& r0, fp2& r1, const fp2& a0, const fp2& a1) const {
        fp v = half_sqr(fp2_select((role >> 1) == 0, a0, a1));
        r0 = gather2(v, 0); r1 = gather2(v, 1);
    }
    // five squares = ten halves on ten lanes: ONE Fp product time (team_lanes8: a whole Fp2 square, two products, on each of five lanes)
    __device__ __forceinline__ void sqr5(fp2& r0, fp2& r1, fp2& r2, fp2& r3, fp2& r4, const fp2& a0, const fp2& a1, const fp2& a2, const fp2& a3, const fp2& a4) const {
        const uint32_t q = role >> 1;
        fp v = half_sqr(fp2_select(q == 0, a0, fp2_select(q == 1, a1, fp2_select(q == 2, a2, fp2_select(q == 3, a3, a4)))));
        r0 = gather2(v, 0); r1 = gather2(v, 1); r2 = gather2(v, 2); r3 = gather2(v, 3); r4 = gather2(v, 4);
    }
    __device__ __forceinline__ void fpmul6(fp (&r)[6], const fp (&a)[6], const fp (&b)[3]) const {
        fp xa = fp_select(role == 0, a[0], fp_select(role == 1, a[1], fp_select(role == 2, a[2], fp_select(role == 3, a[3], fp_select(role == 4, a[4], a[5])))));
        fp xb = fp_select(role < 2, b[0], fp_select(role < 4, b[1], b[2]));
        fp v = fp_mul(xa, xb);
#pragma unroll
        for (int i = 0; i < 6; i++) r[i] = fp_from_role(v, gbase, (uint32_t)i);
    }
};
struct team_lanes8 {
    uint32_t gbase, role;
    // An Fp2 product is two independent dot products, an Fp2 square two independent Fp products: SIX lanes take one HALF each
    // (role r: product r / 2, real part for even r, imaginary for odd), so a round costs one Fp-sized multiplier call
    // (~700 instructions) instead of an Fp2-sized one (~1400): the chain is latency-bound, the other lanes are idle anyway.
    __device__ __forceinline__ fp2 pick3(const fp2& a0, const fp2& a1, const fp2& a2) const {
        const uint32_t q = role >> 1;
        return fp2_select(q == 0, a0, fp2_select(q == 1, a1, a2));
    }
    __device__ __forceinline__ fp2 gather3(const fp& v, uint32_t q) const { return fp2{fp_from_role(v, gbase, 2 * q), fp_from_role(v, gbase, 2 * q + 1)}; }
    __device__ __forceinline__ fp half_mul(const fp2& a, const fp2& b) const {          // real (even role) or imaginary half of a * b
        const bool im = (role & 1) != 0;
        return fp_dot2(a.c0, fp_select(im, b.c1, b.c0), fp_select(im, a.c1, fp_neg(a.c1)), fp_select(im, b.c0, b.c1));
    }
    __device__ __forceinline__ fp half_sqr(const fp2& a) const {                        // (a0 + a1)(a0 - a1)  |  2 a0 a1
        const bool im = (role & 1) != 0;
        return fp_mul(fp_select(im, fp_dbl_nc(a.c0), fp_add_nc(a.c0, a.c1)), fp_select(im, a.c1, fp_sub_nc(a.c0, a.c1)));
    }
    __device__ __forceinline__ void mul3(fp2& r0, fp2& r1, fp2& r2, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1, const fp2& a2, const fp2& b2) const {
        fp v = half_mul(pick3(a0, a1, a2), pick3(b0, b1, b2));
        r0 = gather3(v, 0); r1 = gather3(v, 1); r2 = gather3(v, 2);
    }
    __device__ __forceinline__ void sqr3(fp2& r0, fp2& r1, fp2& r2, const fp2& a0, const fp2& a1, const fp2& a2) const {
        fp v = half_sqr(pick3(a0, a1, a2));
        r0 = gather3(v, 0); r1 = gather3(v, 1); r2 = gather3(v, 2);
    }
    __device__ __forceinline__ fp2 mul1(const fp2& a, const fp2& b) const { return gather3(half_mul(a, b), 0); }
    // four products, eight halves: every lane of the team multiplies
    __device__ __forceinline__ void mul4(fp2& r0, fp2& r1, fp2& r2, fp2& r3, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1, const fp2& a2, const fp2& b2,
                                         const fp2& a3, const fp2& b3) const {
        const uint32_t q = role >> 1;
        fp v = half_mul(fp2_select(q < 2, fp2_select(q == 0, a0, a1), fp2_select(q == 2, a2, a3)), fp2_select(q < 2, fp2_select(q == 0, b0, b1), fp2_select(q == 2, b2, b3)));
        r0 = gather3(v, 0); r1 = gather3(v, 1); r2 = gather3(v, 2); r3 = gather3(v, 3);
    }
    __device__ __forceinline__ void sqr5(fp2& r0, fp2& r1, fp2& r2, fp2& r3, fp2& r4, const fp2& a0, const fp2& a1, const fp2& a2, const fp2& a3, const fp2& a4) const {
        fp2 r = fp2_sqr(fp2_select(role == 0, a0, fp2_select(role == 1, a1, fp2_select(role == 2, a2, fp2_select(role == 3, a3, a4)))));
        r0 = fp2_from_role(r, gbase, 0); r1 = fp2_from_role(r, gbase, 1); r2 = fp2_from_role(r, gbase, 2);
        r3 = fp2_from_role(r, gbase, 3); r4 = fp2_from_role(r, gbase, 4);
    }
    __device__ __forceinline__ void sqr2(fp2& r0, fp2& r1, const fp2& a0, const fp2& a1) const {      // four lanes, one half each
        fp v = half_sqr(fp2_select((role >> 1) == 0, a0, a1));
        r0 = gather3(v, 0); r1 = gather3(v, 1);
    }
    __device__ __forceinline__ void mul2(fp2& r0, fp2& r1, const fp2& a0, const fp2& b0, const fp2& a1, const fp2& b1) const {
        const bool first = (role >> 1) == 0;
        fp v = half_mul(fp2_select(first, a0, a1), fp2_select(first, b0, b1));
        r0 = gather3(v, 0); r1 = gather3(v, 1);
    }
    __device__ __forceinline__ void fpmul6(fp (&r)[6], const fp (&a)[6], const fp (&b)[3]) const {
        fp xa = fp_select(role == 0, a[0], fp_select(role == 1, a[1], fp_select(role == 2, a[2], fp_select(role == 3, a[3], fp_select(role == 4, a[4], a[5])))));
        fp xb = fp_select(role < 2, b[0], fp_select(role < 4, b[1], b[2]));
        fp v = fp_mul(xa, xb);
#pragma unroll
        for (int i = 0; i < 6; i++) r[i] = fp_from_role(v, gbase, (uint32_t)i);
    }
};
__device__ __forceinline__ g2_jac g2_dbl_coop(const g2_jac& p, uint32_t gbase, uint32_t role) { return jac_dbl_team(p, team_lanes8{gbase, role}); }
// clear_cofactor_g2 (h2c.hpp) with the two 63-doubling chains lane-parallel; the chain accumulator stays in registers (inlined
// loop), the base point waits in the registers of the team (every lane holds it anyway)
template <class Team>
__device__ __forceinline__ g2_jac clear_cofactor_g2_team(const g2_jac& p, const Team& team) {
#if defined(BLS_COOP_PARK_LDS) && defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 coop_park_slots[3 * BLS_LDS_SLOT];
    g2_park_lds park{(bls_lds_u32x4*)coop_park_slots};
#else
    g2_park_regs park;
#endif
    auto add = [&](const g2_jac& a, const g2_jac& b) { return jac_add_team(a, b, team); };
    return clear_cofactor_g2_bits(p, park, [&](const g2_jac& a) { return jac_dbl_team(a, team); }, add, add);
}
__device__ __forceinline__ g2_jac clear_cofactor_g2_coop(const g2_jac& p, uint32_t gbase, uint32_t role) { return clear_cofactor_g2_team(p, team_lanes8{gbase, role}); }
__device__ __forceinline__ g2_jac g2_add_coop(const g2_jac& a, const g2_jac& b, uint32_t gbase, uint32_t role) { return jac_add_team(a, b, team_lanes8{gbase, role}); }
// ONE message of any length (fastAggregateVerify / coreVerify shape): latency is all that matters, so a wave works on
// it cooperatively: the two SSWU maps run in roles 0 and 1, the doubling chains of the cofactor clearing spread
// their independent products over roles 0..2.  Every group of 8 lanes does the same work.
__global__ void __launch_bounds__(WAVE) k_hash_one(const uint8_t* __restrict__ msg, uint32_t len, dst_t dst, xmd32_consts xc, uint4* __restrict__ H, size_t stride, size_t slot) {
    const uint32_t role = threadIdx.x & 7u, gbase = threadIdx.x & ~7u;
#ifdef BLS_TAIL_CLOCK
    unsigned long long ts[6];
    ts[0] = __builtin_amdgcn_s_memtime();
#endif
    fp2 u0, u1;
    if (xc.valid && len == 32) {                      // the usual message (a 32-byte signing root): the batch path's 18 compressions with the DST's words prepared
        uint32_t mbe[8];                              // on the host - the byte-wise absorber below costs 0.23 ms for such a message, this form 0.05
#pragma unroll
        for (int j = 0; j < 8; j++) mbe[j] = ((uint32_t)msg[4 * j] << 24) | ((uint32_t)msg[4 * j + 1] << 16) | ((uint32_t)msg[4 * j + 2] << 8) | (uint32_t)msg[4 * j + 3];
        hash_to_field_fp2x2_msg32(u0, u1, mbe, xc);
    } else {
        hash_to_field_fp2x2(u0, u1, msg, len, dst.b, dst.len);
    }
#ifdef BLS_TAIL_CLOCK
    ts[1] = __builtin_amdgcn_s_memtime();
#endif
    g2_jac qs = sswu_g2(fp2_select(role == 1, u1, u0));
#ifdef BLS_TAIL_CLOCK
    ts[2] = __builtin_amdgcn_s_memtime();
#endif
    g2_jac q = iso3_g2(qs);
    g2_jac q0{fp2_from_role(q.x, gbase, 0), fp2_from_role(q.y, gbase, 0), fp2_from_role(q.z, gbase, 0)};
    g2_jac q1{fp2_from_role(q.x, gbase, 1), fp2_from_role(q.y, gbase, 1), fp2_from_role(q.z, gbase, 1)};
#ifdef BLS_TAIL_CLOCK
    ts[3] = __builtin_amdgcn_s_memtime();
#endif
    const team_lanes16 team{threadIdx.x & ~15u, threadIdx.x & 15u};      // the addition and the cofactor chain on quarter products (16 lanes per team)
    g2_jac sum = jac_add_team(q0, q1, team);
#ifdef BLS_TAIL_CLOCK
    ts[4] = __builtin_amdgcn_s_memtime();
#endif
    g2_jac h = clear_cofactor_g2_team(sum, team);
#ifdef BLS_TAIL_CLOCK
    ts[5] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) printf("k_hash_one ticks: hash_to_field %llu  sswu %llu  isogeny+gather %llu  add %llu  cofactor %llu\n", ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4]);
#endif
    if (threadIdx.x == 0 && blockIdx.x == 0) soa_st_g2(H, stride, slot, h);
}
// batch form for batches that would not fill the chip with one lane per message: 8 lanes per message, or 16 (quarter products, team_lanes16) while
// 16 lanes per message still fit the chip's one-per-SIMD wave slots (<= 4 096 messages)
template <int L>
__global__ void __launch_bounds__(WAVE) k_hash_clear_coop(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    const uint32_t role = threadIdx.x & (L - 1), gbase = threadIdx.x & ~(uint32_t)(L - 1);
    uint32_t i = blockIdx.x * (WAVE / L) + (threadIdx.x / L);
    bool live = i < n;
    if (!live) i = 0;
    g2_jac q0 = soa_ld_g2(M, mstride, 2 * (size_t)i), q1 = soa_ld_g2(M, mstride, 2 * (size_t)i + 1);
    g2_jac h;
    if (L == 16) {
        const team_lanes16 team{gbase, role};
        h = clear_cofactor_g2_team(jac_add_team(q0, q1, team), team);
    } else {
        h = clear_cofactor_g2_coop(g2_add_coop(q0, q1, gbase, role), gbase, role);
    }
    if (live && role == 0) soa_st_g2(H, stride, i, h);
}

// G1 arithmetic has a small live set (a Jacobian point is 42 registers): 256 registers, two waves per SIMD, which fill each
// other's issue gaps (non-multiply VALU instructions issue about twice as fast with a second wave on the SIMD)
__global__ void __launch_bounds__(WAVE, 2) k_pkmul(const uint8_t* __restrict__ sets, uint32_t n, const uint64_t* __restrict__ r, uint4* __restrict__ P,
                                                   size_t stride, uint32_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sets + (size_t)i * 320);
    g1_aff pk = ld_g1a_blst(w);
    if (aff_is_inf(pk)) atomicOr(flags, 1u);        // BLST_PK_IS_INFINITY -> update() false
    g1_jac q = jac_mul_u64_w4_body(pk, r[i]);
    soa_st_g1(P, stride, i, q);
}

// ------------------------------------------------------------------------------------------
// k_lines_coop: the Miller lines of FEW pairs (up to 8 per wave): 8 lanes share one pair and split the independent
// products of every doubling step (the 63 of the 68 steps): 5 squarings, then 2 squarings, then 2 products, then the
// 6 Fp products of the line scaling, each group as ONE multiplier call with per-lane operands - about 4 multiplication
// times per step instead of 15.  Same formulas, carries and reductions as miller_dbl_step; the 5 addition steps likewise
// (miller_add_step_team: six rounds).  Used when the pairs would not fill the chip anyway (latency: 2.3 -> ~0.8 ms).
// ------------------------------------------------------------------------------------------
template <int L> struct team_of { typedef team_lanes8 type; };
template <> struct team_of<16> { typedef team_lanes16 type; };
// L = 8 lanes per pair, or 16 (team_lanes16: quarter products, ten half squares in one round) while the pairs leave half the chip's wave slots free
template <int L>
__global__ void __launch_bounds__(WAVE) k_lines_coop(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                     uint4* __restrict__ lines) {
    const uint32_t role = threadIdx.x & (L - 1), gbase = threadIdx.x & ~(uint32_t)(L - 1);
    const typename team_of<L>::type team{gbase, role};
    uint32_t i = blockIdx.x * (WAVE / L) + (threadIdx.x / L);
    bool live = i < count;
    i = first + (live ? i : 0);                                  // idle groups recompute pair `first` (no stores)
    g1_jac pj = soa_ld_g1(P, stride, i);
    g2_jac qj = soa_ld_g2(H, stride, i);
    bool skip = jac_is_inf(pj) | jac_is_inf(qj);
    g1_pre p = g1_precompute(pj);
    g2_proj q = g2_to_proj(qj);
    q = g2_proj{fp2_reduce(q.x), fp2_reduce(q.y), fp2_reduce(q.z)};
    g2_proj t = q;
    int sidx = 0;
    auto sink = [&](const line_t& l0) {
        if (live && role == 0) {
            line_t l = skip ? line_one() : l0;
            uint4* b = lines + (size_t)sidx * 24 * stride;
            soa_st2(b, stride, 0, i, l.l0);
            soa_st2(b, stride, 2, i, l.l1);
            soa_st2(b, stride, 4, i, l.l2);
        }
        sidx++;
    };
#pragma clang loop unroll(disable)
    for (int bit = 62; bit >= 0; bit--) {
        sink(miller_dbl_step_team(t, p, team));
        if ((k::X_ABS >> bit) & 1) sink(miller_add_step_team(t, q, p, team));
    }
}

// lines[s] : 6 fp planes (l0.c0,l0.c1,l1.c0,l1.c1,l2.c0,l2.c1), step-major
__global__ void __launch_bounds__(WAVE) k_lines(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                uint4* __restrict__ lines) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    i += first;
    g1_jac p = soa_ld_g1(P, stride, i);
    g2_jac q = soa_ld_g2(H, stride, i);
    miller_lines(p, q, [&](int s, const line_t& l) {
        uint4* b = lines + (size_t)s * 24 * stride;
        soa_st2(b, stride, 0, i, l.l0);
        soa_st2(b, stride, 2, i, l.l1);
        soa_st2(b, stride, 4, i, l.l2);
    });
}

// The per-lane accumulation loop of k_lineprod as ONE hand-allocated assembly statement (tools/gen_lineprod_asm.py, written to
// build/lineprod_asm.inc by build.sh): f <- line_0, then f <- f * line_j for j = 1 .. rounds - 1 with fp12_mul_by_line_lazy's schoolbook
// product (two six-term Montgomery dot products per coefficient), every value in a fixed register, no scratch, no LDS, no calls;
// lanes whose pair index is past npairs sit out (exec) and keep f = 1.  The statement ends with f stored in st_fp12_int's layout.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_LINEPROD_NOASM)
#include "../build/lineprod_asm.inc"
__device__ __forceinline__ void lineprod_asm(const uint4* step_base, uint32_t stride16, uint32_t npairs, uint32_t first, uint32_t rounds, uint32_t* out,
                                             uint32_t lds_buf) {
    asm volatile(BLS_LINEPROD_ASM_BODY : : "s"(step_base), "s"(stride16), "s"(npairs), "s"(first), "s"(rounds), "s"(out), "s"(lds_buf) : BLS_LINEPROD_ASM_CLOBBERS);
}
#endif
// grid (N_LINES, nblk): block b of step s multiplies lines of pairs b*64*m .. (b+1)*64*m
__global__ void __launch_bounds__(WAVE) k_lineprod(const uint4* __restrict__ lines, uint32_t npairs, size_t stride, uint32_t m,
                                                   uint32_t* __restrict__ part, uint32_t nblk, int per_lane) {
    uint32_t s = blockIdx.x, b = blockIdx.y;
    const uint4* base = lines + (size_t)s * 24 * stride;
    size_t first = (size_t)b * WAVE * m;
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 line_slots[4 * BLS_LDS_SLOT];              // 28 KB: with the 7 KB hand-over slot 35 of the 40 KB a wave may use
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_LINEPROD_NOASM)
    if (per_lane == 1) {                              // wave-uniform; 2 = the compiled per-lane path below (byte offsets beyond 32 bits)
        uint32_t rounds = 0;
        if (first < npairs) {
            size_t left = (npairs - first + WAVE - 1) / WAVE;
            rounds = left < m ? (uint32_t)left : m;
        }
        // the next line travels HBM -> LDS (LDS-DMA, 24 rows of 1 KiB) while the current one is multiplied: the same 28 KiB the compiled path parks its line in
        lineprod_asm(base, (uint32_t)(stride * 16), npairs, (uint32_t)first, rounds, part + ((size_t)s * nblk + b) * WAVE * F12W,
                     (uint32_t)(uintptr_t)(bls_lds_u32x4*)line_slots);
        return;
    }
#endif
    // Straight-line accumulation (no "have a value yet" flag, no per-lane conditional update: those made the
    // compiler keep f in scratch memory across iterations).  Lanes past the end multiply by the line 1; a round
    // with no valid lane at all ends the loop (wave-uniform test).
    auto ld_line = [&](size_t j0) {
        size_t i = j0 + threadIdx.x;
        bool v = i < npairs;
        size_t ia = v ? i : j0;
        line_t l{soa_ld2(base, stride, 0, ia), soa_ld2(base, stride, 2, ia), soa_ld2(base, stride, 4, ia)};
        return line_t{fp2_select(v, l.l0, fp2_one()), fp2_select(v, l.l1, fp2_zero()), fp2_select(v, l.l2, fp2_zero())};
    };
    fp12 f = fp12_one();
#if defined(__HIP_DEVICE_COMPILE__)
    line_ops_lds lops{(bls_lds_u32x4*)line_slots};
#else
    line_ops_lds lops{};
#endif
    if (first < npairs) {
        f = fp12_from_line(ld_line(first));
#pragma clang loop unroll(disable)
        for (uint32_t j = 1; j < m; j++) {
            size_t j0 = first + (size_t)j * WAVE;
            if (j0 >= npairs) break;
            lops.park(ld_line(j0));                  // the compiled path keeps the compact Karatsuba form on the shared multiplier bodies
            f = fp12_mul_by_line_ops(f, lops);
        }
    }
    f = fp12_reduce(f);
    if (per_lane) {
        // large batches: every lane hands its partial product to k_lineprod2 (64 x nblk partials per step).  The
        // in-wave shuffle tree below costs six Fp12 products of wave time for 63 lane-products of work; done by
        // k_lineprod2's 68 waves instead (15 sequential products per lane + one tree) it is ~4x less wave time.
        st_fp12_int(part + (((size_t)s * nblk + b) * WAVE + threadIdx.x) * F12W, f);
        return;
    }
    // lanes past the last pair of this range hold 1: skip the tree levels that would only fold ones (wave-uniform)
    size_t live = first < npairs ? npairs - first : 0;
    int top = 32;
    while (top >= 1 && (size_t)top >= live) top >>= 1;
    for (int d = top; d >= 1; d >>= 1) {
        fp12 o = shfl_down_struct(f, d);
        f = fp12_mul(f, o);
    }
    if (threadIdx.x == 0) st_fp12_int(part + ((size_t)s * nblk + b) * F12W, f);
}

// per step: product of the nblk partials of k_lineprod -> L_s
__global__ void __launch_bounds__(WAVE) k_lineprod2(const uint32_t* __restrict__ part, uint32_t nblk, uint32_t* __restrict__ L) {
    uint32_t s = blockIdx.x;
    fp12 f = fp12_one();
    for (uint32_t j = threadIdx.x; j < nblk; j += WAVE) {
        fp12 o = ld_fp12_int(part + ((size_t)s * nblk + j) * F12W);
        f = j < WAVE ? o : fp12_mul(f, o);
    }
    int top = 32;                                   // lanes >= nblk hold 1: skip the tree levels that only fold ones
    while (top >= 1 && (uint32_t)top >= nblk) top >>= 1;
    for (int d = top; d >= 1; d >>= 1) {
        fp12 o = shfl_down_struct(f, d);
        f = fp12_mul(f, o);
    }
    if (threadIdx.x == 0) st_fp12_int(L + (size_t)s * F12W, f);
}

// ------------------------------------------------------------------------------------------
// Wave-cooperative Fp12 engine for the per-batch serial tail (Horner over the 68 step products,
// shard merge, final exponentiation).  One wave; Fp12 values live in LDS in the flat basis
// Fp2[w]/(w^6 - xi) (tower slots c0.(a0,a1,a2), c1.(a0,a1,a2) = w^0,2,4 / w^1,3,5).
// A product is 36 lanes x one Fp2 multiplication (a_i * b_j) + 12 lanes x one 6-term column sum,
// i.e. ~1.3 Fp2-mul latencies instead of 18 on a single lane.
// ------------------------------------------------------------------------------------------
constexpr int C12_NREG = 8;
// Threads of the engine's workgroup (the kernels read blockDim.x).  Three waves run the 108 products of an Fp12 multiplication
// and the 168 items of its second phase in ONE round each: the latency-mode launch (two waves: 8 % slower).  Throughput mode
// launches two waves: with the chip saturated by 512-register waves a workgroup waits until enough SIMDs of ONE CU have
// drained, at the head of a hardware queue that other callers' streams share (three waves: +3.6 % per pipelined batch).
#ifdef BLS_C12_ROW
constexpr int TAIL_THREADS = 192, TAIL_THREADS_TP = 192;          // the row engine: 12 rows of 16 lanes
#else
constexpr int TAIL_THREADS = 192, TAIL_THREADS_TP = 192;          // >= 168: one limb item per thread in phase 2a (and >= 108 products in phase 1)
#endif          // the row engine: 12 rows of 16 lanes (the Karatsuba engine needed >= 108)
struct c12_lds {
    fp2 r[C12_NREG][6];
    c12_work w;
    fp2 frob[6];
    fp frob2[6];
    uint32_t steps[N_LINES * 6 * 2 * FP_N];     // the 68 step products, flat basis (46 KB): loaded once, no global load per Horner step
};

#ifdef BLS_TAIL_CLOCK
#define C12_T0 unsigned long long last_ = __builtin_amdgcn_s_memtime()
#define C12_STAMP(i) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) S.w.prof[i] += now_ - last_; last_ = now_; } while (0)
#else
#define C12_T0 do { } while (0)
#define C12_STAMP(i) do { } while (0)
#endif
#ifndef BLS_C12_ROW
// Phase 1, schoolbook (c12.hpp, c12s_*): thread q < 144 (84 for a square) forms ONE Fp product of two operands it picks by address - no operand sums, no selects,
// the three waves equally loaded (the Karatsuba form, three products per pair with the kind uniform per wave, left the third wave 28 LDS reads and 28
// additions behind the other two: products 2.8 k cycles + 0.33 k of waiting at the barrier).
template <int NITEMS, bool SQR, class LDS>
__device__ __forceinline__ void c12_products(LDS& S, int a, int b) {
    const int q = (int)threadIdx.x;
    if (q < NITEMS) S.w.prod[q] = c12s_product(S.r[a], S.r[b], q, SQR);
}
// Phase 2 on rows (c12.hpp): thread (c, l) = (t / 16, t % 16) keeps its limb in a register from the 18-term sum to the stored result; carries travel by
// DPP row shifts, the quotient of the partial reduction comes from lane 13 of the row (v_readlane per row of the wave).  One barrier less per product and
// ~60 instructions on every lane instead of ~280 on twelve.
__device__ __forceinline__ int32_t c12_shr1(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }      // lane l <- lane l - 1 of its 16-lane row, 0 into lane 0
template <bool SQR, class LDS>
__device__ __forceinline__ void c12_phase2_rows(LDS& S, int d) {
    const int t = threadIdx.x, c = t >> 4, l = t & 15;
    const bool live = l < FP_N;
    const int ll = live ? l : 0;
    const int32_t pl = S.w.pl[l];                                          // issued with the other LDS reads, used at the end
    const c12_lc x = c12_split(c12s_limb_sum(S.w, c, ll, SQR), ll);
    const int32_t limb = x.lo + c12_shr1(live ? x.car : 0);
    const int32_t qv = c12_quotient(limb + c12_shr1(limb >> 28));          // right in lane 13 of the row
    const int32_t q0 = __builtin_amdgcn_readlane(qv, 13), q1 = __builtin_amdgcn_readlane(qv, 29), q2 = __builtin_amdgcn_readlane(qv, 45), q3 = __builtin_amdgcn_readlane(qv, 61);
    const int r = (t >> 4) & 3;
    const int32_t q = r == 0 ? q0 : (r == 1 ? q1 : (r == 2 ? q2 : q3));
    const c12_lc y = c12_split(c12_sub_qp_v(limb, q, pl), ll);
    const int32_t out = y.lo + c12_shr1(live ? y.car : 0);
    if (live) {
        fp2& dst = S.r[d][c >> 1];
        ((c & 1) ? dst.c1 : dst.c0).l[l] = (uint32_t)out;
    }
}
template <class LDS>
__device__ __noinline__ void c12_mul(LDS& S, int d, int a, int b) {
    C12_T0;
    c12_products<144, false>(S, a, b);
    C12_STAMP(0);
    __syncthreads();
    C12_STAMP(1);
    c12_phase2_rows<false>(S, d);
    C12_STAMP(2);
    __syncthreads();
    C12_STAMP(3);
}
// d = a^2: only the 21 pairs i <= j are formed, 63 Fp products
__device__ __noinline__ void c12_sqr(c12_lds& S, int d, int a) {
    C12_T0;
    c12_products<84, true>(S, a, a);
    C12_STAMP(0);
    __syncthreads();
    C12_STAMP(1);
    c12_phase2_rows<true>(S, d);
    C12_STAMP(2);
    __syncthreads();
    C12_STAMP(3);
}
#else
// Row engine (c12.hpp, c12r_*; -DBLS_C12_ROW): thread (row = t / 16, q = t % 16) of the 192; d may be a or b (every operand is read before the first
// barrier, the results are written after it).  The row sum runs on DPP row shifts: lane 15 of a row ends with the sum of its 16 lanes.
// NOT the default: ~1 200 instructions per product against the Karatsuba engine's ~1 650 on the critical lane, and its best launches are 15 - 20 %
// faster (k_tail Horner 0.37 ms against 0.43, final exponentiation 1.07 against 1.3), but the SAME launch takes 1.0x / 1.24x / 1.4x / 1.55x that
// depending on the CU it lands on (stable per CU within a process, different CUs on different boxes; clock constant at 2.39 - 2.44 GHz; every
// instruction class alone is uniform over the CUs: tools/ubench_cu.hip), so its average is 5 - 10 % SLOWER (profiles/r04_ab/row_engine.txt).
#ifdef BLS_ROW_SHFL
#define BLS_ROW_GET(VV, CTRL) (((int)(threadIdx.x & 15) >= ((CTRL) & 15)) ? (uint32_t)__shfl_up((int)(VV), (CTRL) & 15, 16) : 0u)
#else
#define BLS_ROW_GET(VV, CTRL) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(VV), CTRL, 0xf, 0xf, true))
#endif
__device__ __forceinline__ void c12r_row_sum(c12r_limbs& v) {
#define BLS_ROW_STEP(CTRL)                                                                                                        \
    _Pragma("unroll") for (int l = 0; l < 2 * FP_N; l++) v.t[l] += BLS_ROW_GET(v.t[l], CTRL);
    BLS_ROW_STEP(0x111)      // row_shr:1
    BLS_ROW_STEP(0x112)      // row_shr:2
    BLS_ROW_STEP(0x114)      // row_shr:4
    BLS_ROW_STEP(0x118)      // row_shr:8
#undef BLS_ROW_STEP
}
template <class LDS>
__device__ __noinline__ void c12_mul(LDS& S, int d, int a, int b) {
    const int t = threadIdx.x, row = (t >> 4) < 12 ? (t >> 4) : 0, q = (t >> 4) < 12 ? (t & 15) : 12;      // rows 12.. of a wider block: idle terms
    c12r_limbs v;
    c12r_term(S.r[a], S.r[b], row, q, v);
    __syncthreads();
    c12r_row_sum(v);
    if (q == 15) {
        fp r = c12r_reduce(v);
        if (row & 1) S.r[d][row >> 1].c1 = r; else S.r[d][row >> 1].c0 = r;
    }
    __syncthreads();
}
__device__ __forceinline__ void c12_sqr(c12_lds& S, int d, int a) { c12_mul(S, d, a, a); }
#endif
__device__ __forceinline__ void c12_copy(c12_lds& S, int d, int a) {
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][lane] = S.r[a][lane];
    __syncthreads();
}
__device__ __forceinline__ void c12_conj(c12_lds& S, int d, int a) {    // w -> -w
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][lane] = (lane & 1) ? fp2_neg(S.r[a][lane]) : S.r[a][lane];
    __syncthreads();
}
__device__ __forceinline__ void c12_frob(c12_lds& S, int d, int a) {
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][lane] = fp2_mul(fp2_conj(S.r[a][lane]), S.frob[lane]);
    __syncthreads();
}
__device__ __forceinline__ void c12_frob2(c12_lds& S, int d, int a) {
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][lane] = fp2_mul_fp(S.r[a][lane], S.frob2[lane]);
    __syncthreads();
}
template <class LDS>
__device__ __forceinline__ void c12_load(LDS& S, int d, const uint32_t* g) {   // blst_fp12 image (576 B)
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][c12_flat_of_tower(lane)] = fp2{ld_fp_blst(g + 24 * lane), ld_fp_blst(g + 24 * lane + 12)};
    __syncthreads();
}
// step product s (internal Fp12 layout: tower order, FPW words per Fp) from the LDS copy
__device__ __forceinline__ void c12_load_step(c12_lds& S, int d, int s) {
    int lane = threadIdx.x;
    if (lane < 12) {
        const uint32_t* w = S.steps + ((size_t)s * 12 + lane) * FP_N;
        fp v;
#pragma unroll
        for (int i = 0; i < FP_N; i++) v.l[i] = w[i];
        fp2& dst = S.r[d][c12_flat_of_tower(lane >> 1)];
        if (lane & 1) dst.c1 = v; else dst.c0 = v;
    }
    __syncthreads();
}
template <class LDS>
__device__ __forceinline__ void c12_store(const LDS& S, int a, uint32_t* g) {
    int lane = threadIdx.x;
    if (lane < 6) {
        const fp2& v = S.r[a][c12_flat_of_tower(lane)];
        st_fp_blst(g + 24 * lane, v.c0);
        st_fp_blst(g + 24 * lane + 12, v.c1);
    }
    __syncthreads();
}
__device__ __forceinline__ void c12_set_one(c12_lds& S, int d) {
    int lane = threadIdx.x;
    if (lane < 6) S.r[d][lane] = lane == 0 ? fp2_one() : fp2_zero();
    __syncthreads();
}
// internal Fp12 layout (tower order, FPW words per Fp) <-> engine register
template <class LDS>
__device__ __forceinline__ void c12_load_int(LDS& S, int d, const uint32_t* g) {
    int lane = threadIdx.x;
    if (lane < 12) {
        fp v = ld_fp_int(g + lane * FPW);
        fp2& dst = S.r[d][c12_flat_of_tower(lane >> 1)];
        if (lane & 1) dst.c1 = v; else dst.c0 = v;
    }
    __syncthreads();
}
template <class LDS>
__device__ __forceinline__ void c12_store_int(const LDS& S, int a, uint32_t* g) {
    int lane = threadIdx.x;
    if (lane < 12) {
        const fp2& v = S.r[a][c12_flat_of_tower(lane >> 1)];
        st_fp_int(g + lane * FPW, (lane & 1) ? v.c1 : v.c0);
    }
    __syncthreads();
}
// Latency mode: the per-lane partial products of k_lineprod folded on the engine.  grid (N_LINES, nb): block (s, b) multiplies
// `per` consecutive partials of step s (the last block: last_count) -> dst[s * nb + b]; two levels of about sqrt(count)
// dependent engine products (~4.5 us each) where the in-wave shuffle tree (6 whole Fp12 multiplications per lane) plus a
// per-lane fold took ~0.8 ms.
struct fold_lds {
    fp2 r[2][6];
    c12_work w;
};
__global__ void __launch_bounds__(TAIL_THREADS) k_fold(const uint32_t* __restrict__ src, uint32_t count, uint32_t per, uint32_t last_count,
                                                       uint32_t* __restrict__ dst) {
    __shared__ fold_lds S;
    c12_fill_p(S.w, (int)threadIdx.x);               // visible after the barrier of the first load below
    const uint32_t s = blockIdx.x, b = blockIdx.y, nb = gridDim.y;
    const uint32_t lo = b * per, cnt = b + 1 == nb ? last_count : per;
    const uint32_t* g = src + ((size_t)s * count + lo) * F12W;
    c12_load_int(S, 0, g);
    for (uint32_t j = 1; j < cnt; j++) {
        c12_load_int(S, 1, g + (size_t)j * F12W);
        c12_mul(S, 0, 0, 1);
    }
    c12_store_int(S, 0, dst + ((size_t)s * nb + b) * F12W);
}

// Committed pairing states (blst_fp12 images, 144 words each) on the engine: states[dst] = states[a] * states[b], or a copy of
// states[a] when b < 0.  blst_pairing_merge (blst_abi.nim:508) between the slices of a batch that is larger than the context's
// capacity: every slice commits its own state, the running product lives in slot 1.
__global__ void __launch_bounds__(TAIL_THREADS) k_state_mul(uint32_t* __restrict__ states, int dst, int a, int b) {
    __shared__ fold_lds S;
    c12_fill_p(S.w, (int)threadIdx.x);
    c12_load(S, 0, states + (size_t)a * 144);
    if (b >= 0) {
        c12_load(S, 1, states + (size_t)b * 144);
        c12_mul(S, 0, 0, 1);
    }
    c12_store(S, 0, states + (size_t)dst * 144);
}

// d = a^x (x < 0, a cyclotomic): square-and-multiply over |x|, then conjugate.  tmp != a.
__device__ __noinline__ void c12_cyc_exp_x(c12_lds& S, int d, int a, int tmp) {
    c12_copy(S, tmp, a);
    for (int bit = 62; bit >= 0; bit--) {
        c12_sqr(S, tmp, tmp);
        if ((k::X_ABS >> bit) & 1) c12_mul(S, tmp, tmp, a);
    }
    c12_conj(S, d, tmp);
}
// d = 1 / a = conj_6(a) / (a conj_6(a)): the norm to Fp6 and the last product run on the lane-parallel engine; the Fp6
// inversion (one Fp inversion inside: the only long single-lane step of the tail) on lane 0.  t1, t2: scratch registers.
__device__ __noinline__ void c12_inv(c12_lds& S, int d, int a, int t1, int t2) {
    c12_conj(S, t1, a);                    // (c0, -c1)
    c12_mul(S, t2, a, t1);                 // c0^2 - v c1^2: an Fp6 element, odd powers of w are zero
    if (threadIdx.x == 0) {
        fp6 n{S.r[t2][0], S.r[t2][2], S.r[t2][4]};
        fp6 ni = fp6_inv(n);
        S.r[t2][0] = ni.a0; S.r[t2][2] = ni.a1; S.r[t2][4] = ni.a2;
        S.r[t2][1] = fp2_zero(); S.r[t2][3] = fp2_zero(); S.r[t2][5] = fp2_zero();
    }
    __syncthreads();
    c12_mul(S, d, t1, t2);
}

// One block of two waves.  mode bit 0: Horner-combine the 68 step products L -> state slot 0 (Miller value);
// bit 1: multiply the kk states and run the final exponentiation -> gt_out, verdict.
// sstride: distance in words between the kk states (144 = packed blst_fp12 images); blob != 0: every state is followed by
// its shard's ok word (1 = no update failed), and the verdict also requires all of them.
__global__ void __launch_bounds__(TAIL_THREADS) k_tail(const uint32_t* __restrict__ L, uint32_t* __restrict__ states, uint32_t kk, int mode,
                                                       uint32_t* __restrict__ gt_out, uint32_t* __restrict__ verdict, uint32_t sstride, int blob) {
    __shared__ c12_lds S;
    int lane = threadIdx.x;
    c12_fill_p(S.w, lane);                           // the row phase's table of p's limbs (visible after the barrier below)
#ifdef BLS_TAIL_CLOCK
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x < 8) S.w.prof[threadIdx.x] = 0;
#endif
    if (lane == 0) {
        S.frob[0] = fp2_one(); S.frob[1] = fp2_from_const(k::FROB_G1); S.frob[2] = fp2_from_const(k::FROB_G2);
        S.frob[3] = fp2_from_const(k::FROB_G3); S.frob[4] = fp2_from_const(k::FROB_G4); S.frob[5] = fp2_from_const(k::FROB_G5);
        S.frob2[0] = fp_one(); S.frob2[1] = fp_from_const(k::FROB2_G1); S.frob2[2] = fp_from_const(k::FROB2_G2);
        S.frob2[3] = fp_from_const(k::FROB2_G3); S.frob2[4] = fp_from_const(k::FROB2_G4); S.frob2[5] = fp_from_const(k::FROB2_G5);
    }
    if (mode & 1) {
        // all 68 step products into LDS, dropping the two pad words of every Fp
        for (int e = lane; e < N_LINES * 12 * FP_N; e += (int)blockDim.x) S.steps[e] = L[(size_t)(e / FP_N) * FPW + (e % FP_N)];
    }
    __syncthreads();
    enum { F = 0, T = 1, A = 2, B = 3, C = 4, X1 = 5, X2 = 6, X3 = 7 };
    if (mode & 1) {
        // f = conj( Horner_s (f^2 [doubling steps] * L_s) )
        c12_set_one(S, F);
        int s = 0;
        for (int bit = 62; bit >= 0; bit--) {
            c12_sqr(S, F, F);
            c12_load_step(S, X1, s++);
            c12_mul(S, F, F, X1);
            if ((k::X_ABS >> bit) & 1) {
                c12_load_step(S, X1, s++);
                c12_mul(S, F, F, X1);
            }
        }
        c12_conj(S, F, F);
        c12_store(S, F, states);
    }
    if (mode & 2) {
        c12_load(S, F, states);
        for (uint32_t i = 1; i < kk; i++) {
            c12_load(S, X1, states + (size_t)i * sstride);
            c12_mul(S, F, F, X1);
        }
        // easy part: t = conj(f)/f ; t = frob2(t) * t
        c12_inv(S, X1, F, X2, X3);
        c12_conj(S, X2, F);
        c12_mul(S, T, X2, X1);
        c12_frob2(S, X1, T);
        c12_mul(S, T, X1, T);
        // hard part: 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3
        c12_cyc_exp_x(S, X1, T, X3);            // t^x
        c12_conj(S, X2, T);
        c12_mul(S, A, X1, X2);                  // a = t^(x-1)
        c12_cyc_exp_x(S, X1, A, X3);
        c12_conj(S, X2, A);
        c12_mul(S, A, X1, X2);                  // a = t^((x-1)^2)
        c12_cyc_exp_x(S, X1, A, X3);
        c12_frob(S, X2, A);
        c12_mul(S, B, X1, X2);                  // b = a^(x+p)
        c12_cyc_exp_x(S, X1, B, X3);
        c12_cyc_exp_x(S, X2, X1, X3);           // b^(x^2)
        c12_frob2(S, X1, B);
        c12_mul(S, C, X2, X1);
        c12_conj(S, X1, B);
        c12_mul(S, C, C, X1);                   // c = b^(x^2+p^2-1)
        c12_sqr(S, X1, T);
        c12_mul(S, X1, X1, T);                  // t^3
        c12_mul(S, C, C, X1);
        c12_store(S, C, gt_out);
        if (lane == 0) {
            bool one = fp2_eq(S.r[C][0], fp2_one());
            for (int i = 1; i < 6; i++) one = one & fp2_is_zero(S.r[C][i]);
            if (blob)
                for (uint32_t i = 0; i < kk; i++) one = one & (states[(size_t)i * sstride + 144] == 1u);
            *verdict = one ? 1u : 0u;
        }
    }
#ifdef BLS_TAIL_CLOCK
    if ((lane & 63) == 0) {
        uint64_t dr = __builtin_amdgcn_s_memrealtime() - rt0;
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t ldsa;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(ldsa));
#ifndef BLS_C12_ROW
        if (lane == 0) printf("   engine phases, ticks of thread 0 (whole kernel %llu): products %llu  barrier %llu  limb sums + reduction %llu  barrier %llu\n",
                              (unsigned long long)(__builtin_amdgcn_s_memtime() - clk0), S.w.prof[0], S.w.prof[1], S.w.prof[2], S.w.prof[3]);
#endif
        printf("k_tail mode %d wave %d: %.3f ms simd %u wave_slot %u cu %u se %u raw %x lds_base %u lds_size %u (granules; raw %x)\n", mode, lane >> 6, (double)dr / 1e5, (hwid >> 4) & 3, hwid & 15,
               (hwid >> 8) & 15, (hwid >> 13) & 7, hwid, ldsa & 0xff, (ldsa >> 12) & 0x1ff, ldsa);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// G1 point-sum reduction (aggregateAll, blst_min_pubkey_sig_core.nim:179-195; the streaming part
// of fastAggregateVerify, bls_sig_min_pubkey.nim:234-258): 96 B in per ~11 Fp multiplications.
// Lane l of block b sums points (b*64 + l) + j*64*gridDim.x, j < m, with mixed additions, then the
// wave folds its 64 partial sums with shuffles.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE, 2) k_g1_sum(const uint8_t* __restrict__ pts, uint32_t n, uint32_t m, uint32_t* __restrict__ part) {
    uint32_t lane0 = blockIdx.x * WAVE + threadIdx.x, strideL = gridDim.x * WAVE;
    g1_jac acc = jac_inf<fp>();
    for (uint32_t j = 0; j < m; j++) {
        uint32_t i = lane0 + j * strideL;
        if (i < n) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(pts + (size_t)i * 96);
            g1_aff q = ld_g1a_blst(w);
            acc = jac_add_aff(acc, q);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        g1_jac o = shfl_down_struct(acc, d);
        acc = jac_add_body(acc, o);
    }
    if (threadIdx.x == 0) {
        st_g1_int(part + (size_t)blockIdx.x * G1W, acc);
    }
}
__global__ void __launch_bounds__(WAVE) k_g1_sum2(const uint32_t* __restrict__ part, uint32_t nparts, uint32_t* __restrict__ out) {
    g1_jac acc = jac_inf<fp>();
    for (uint32_t j = threadIdx.x; j < nparts; j += WAVE) {
        acc = jac_add_body(acc, ld_g1_int(part + (size_t)j * G1W));
    }
    for (int d = 32; d >= 1; d >>= 1) {
        g1_jac o = shfl_down_struct(acc, d);
        acc = jac_add_body(acc, o);
    }
    if (threadIdx.x == 0) st_g1_blst(out, acc);        // blst_p1 image
}
// pairs of coreVerifyNoGroupCheck (core :269-297): slot 0 = (aggregate pk, H(msg)) [H written by k_hash_one],
// slot 1 = (-G1, signature).  Aggregate at infinity -> BLST_PK_IS_INFINITY flag.
__global__ void k_fav_setup(const uint32_t* __restrict__ agg, const uint32_t* __restrict__ sig, uint4* __restrict__ H, uint4* __restrict__ P,
                            size_t stride, uint32_t* __restrict__ flags) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    g1_jac a = ld_g1_blst(agg);
    if (jac_is_inf(a)) atomicOr(flags, 1u);
    soa_st_g1(P, stride, 0, a);
    soa_st_g1(P, stride, 1, g1_jac{fp_from_const(k::G1_X), fp_from_const(k::G1_NEG_Y), fp_one()});
    g2_aff sg = ld_g2a_blst(sig);
    soa_st_g2(H, stride, 1, jac_from_aff(sg));
}

// ------------------------------------------------------------------------------------------
// Counting-sort helpers shared by the signature-side bucket fold (unsigned digits of the 64-bit blinding scalars) and,
// for the scan, by the Pippenger kernels further down.
// ------------------------------------------------------------------------------------------
// Window w covers bits [off_w, off_w + len_w): the nbits are split into nwin windows whose widths differ by
// at most one bit (wbase + 1 for the first wrem windows, wbase after) so that no window is short and
// concentrates the points into a few buckets.
struct msm_win {
    uint32_t nwin, wbase, wrem;
};
__device__ __forceinline__ uint32_t msm_win_off(const msm_win& W, uint32_t w) {
    return w < W.wrem ? w * (W.wbase + 1) : W.wrem * (W.wbase + 1) + (w - W.wrem) * W.wbase;
}
// scalars are little-endian, sbytes bytes each (32: blst_scalar images; 8: the u64 blinding scalars)
__device__ __forceinline__ uint32_t msm_digit(const uint8_t* __restrict__ sc, size_t i, uint32_t w, const msm_win& W, uint32_t sbytes) {
    uint32_t bit0 = msm_win_off(W, w);
    uint32_t len = w < W.wrem ? W.wbase + 1 : W.wbase;
    const uint8_t* p = sc + i * sbytes;
    uint32_t byte0 = bit0 >> 3;
    uint64_t v = 0;
    for (uint32_t j = 0; j < 4 && byte0 + j < sbytes; j++) v |= (uint64_t)p[byte0 + j] << (8 * j);
    return (uint32_t)(v >> (bit0 & 7)) & ((1u << len) - 1u);
}
__global__ void __launch_bounds__(WAVE) k_msm_hist(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, msm_win W, uint32_t c, uint32_t* __restrict__ hist) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x, w = blockIdx.y;       // lane per (point, window)
    if (i >= n) return;
    uint32_t d = msm_digit(sc, i, w, W, sbytes);
    if (d) atomicAdd(&hist[((size_t)w << c) | d], 1u);
}
// one wave per window
__global__ void __launch_bounds__(WAVE) k_msm_scan(const uint32_t* __restrict__ hist, uint32_t c, uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor) {
    uint32_t w = blockIdx.x, nb = 1u << c, per = (nb + WAVE - 1) / WAVE;
    uint32_t lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    const uint32_t* h = hist + ((size_t)w << c);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += h[b];
    uint32_t incl = sum;
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, WAVE);
        if ((int)threadIdx.x >= d) incl += o;
    }
    uint32_t run = incl - sum;
    for (uint32_t b = lo; b < hi; b++) {
        offs[((size_t)w << c) | b] = run;
        cursor[((size_t)w << c) | b] = run;
        run += h[b];
    }
}
__global__ void __launch_bounds__(WAVE) k_msm_scatter(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, msm_win W, uint32_t c,
                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x, w = blockIdx.y;
    if (i >= n) return;
    uint32_t d = msm_digit(sc, i, w, W, sbytes);
    if (d) {
        uint32_t pos = atomicAdd(&cursor[((size_t)w << c) | d], 1u);
        sorted[(size_t)w * n + pos] = i;
    }
}
// ------------------------------------------------------------------------------------------
// Pippenger for G1 AND G2 (blst_p1s_mult_pippenger / blst_p2s_mult_pippenger, blst_abi.nim:336-362), templated on the
// coordinate field, with SIGNED window digits: k' = k + H, H = sum_w 2^(off_w + len_w - 1), makes every window digit of k' minus
// its half a digit in [-2^(len-1), 2^(len-1) - 1]; a negative digit adds the negated point (free: -y), so a window needs
// 2^(len-1) buckets instead of 2^len and the running-sum reduction halves.  The windows cover nbits + 1 bits (the extra top bit
// of k is zero): the TOP window is left unbiased and absorbs the carry of the bias, its digit stays within 0 .. 2^(len-1).
// Bucket b of a window holds the points with |digit| = b + 1.
//   k_pip_hist / k_msm_scan / k_pip_scatter   counting sort of (point, sign) by window and |digit|; lane per (point, window)
//   k_msm_order_*                             buckets ordered by load so that a wave's lanes do equal work
//   k_pip_bucket    lane per bucket: sum of its signed points (mixed additions)
//   k_pip_segred    lane per segment of 16 buckets: running sums -> sum (b + 1) B_b of the segment
//   k_pip_winpart / k_pip_winsum    per window: sum of the segment values, times 2^(off_w) (lane-parallel doubling chain)
//   k_pip_final     sum over the windows -> blst_p1 / blst_p2 image
// ------------------------------------------------------------------------------------------
struct pip_win {
    uint32_t nwin, wbase, wrem, nbits;      // nwin windows over nbits + 1 bits (widths differ by at most one bit)
    uint32_t cbk;                           // bucket index bits: 2^cbk buckets per window, cbk = widest window - 1
    uint32_t H[9];                          // the bias: 2^(len - 1) at every window but the top one
};
__device__ __forceinline__ uint32_t pip_off(const pip_win& W, uint32_t w) {
    return w < W.wrem ? w * (W.wbase + 1) : W.wrem * (W.wbase + 1) + (w - W.wrem) * W.wbase;
}
// Signed digit of window w of scalar i: k' = (k mod 2^nbits) + H (9 words; scalars little-endian, sbytes each), the len bits
// of k' at the window's offset, minus the half (the top window is unbiased: 0 .. 2^(len-1), carry of the bias included).
// The scalar is fetched with two 16-byte loads when it can be (blst_scalar arrays: sbytes = 32, 16-byte aligned) - one load
// per word behind its own bounds test leaves a lane with eight dependent memory latencies; the two words of k' the window
// needs are picked while the carry runs (w is uniform: no indexed register array).
__device__ __forceinline__ int32_t pip_digit(const uint8_t* __restrict__ sc, size_t i, const pip_win& W, uint32_t sbytes, uint32_t w) {
    const uint8_t* p = sc + i * sbytes;
    uint32_t k[8];
    if (sbytes == 32 && (((uintptr_t)sc) & 15) == 0) {
        const uint4* pv = reinterpret_cast<const uint4*>(p);
        uint4 a = pv[0], b = pv[1];
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    } else if ((sbytes & 3) == 0 && (((uintptr_t)sc) & 3) == 0) {
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = (uint32_t)(4 * j) < sbytes ? pw[j] : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = 0;
            for (int q = 0; q < 4; q++)
                if ((uint32_t)(4 * j + q) < sbytes) v |= (uint32_t)p[4 * j + q] << (8 * q);
            k[j] = v;
        }
    }
    const uint32_t len = w < W.wrem ? W.wbase + 1 : W.wbase, bit0 = pip_off(W, w), wi = bit0 >> 5, sh = bit0 & 31;
    uint32_t carry = 0, lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t b0 = (uint32_t)(32 * j), v = k[j];
        if (W.nbits <= b0) v = 0;
        else if (W.nbits < b0 + 32) v &= (1u << (W.nbits - b0)) - 1u;
        uint64_t t = (uint64_t)v + W.H[j] + carry;
        carry = (uint32_t)(t >> 32);
        if ((uint32_t)j == wi) lo = (uint32_t)t;
        if ((uint32_t)j == wi + 1) hi = (uint32_t)t;
    }
    uint32_t k8 = W.H[8] + carry;
    if (wi == 8) lo = k8;
    if (wi == 7) hi = k8;
    int32_t raw = (int32_t)((uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & ((1u << len) - 1u));       // len <= 31
    return w + 1 == W.nwin ? raw : raw - (int32_t)(1u << (len - 1));
}
// lane per (point, window)
__global__ void __launch_bounds__(WAVE) k_pip_hist(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, pip_win W, uint32_t w0, uint32_t* __restrict__ hist) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x, w = w0 + blockIdx.y;
    if (i >= n) return;
    int32_t d = pip_digit(sc, i, W, sbytes, w);
    if (d) atomicAdd(&hist[((size_t)w << W.cbk) + (uint32_t)((d < 0 ? -d : d) - 1)], 1u);
}
__global__ void __launch_bounds__(WAVE) k_pip_scatter(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, pip_win W, uint32_t w0, uint32_t* __restrict__ cursor,
                                                      uint32_t* __restrict__ sorted) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x, w = w0 + blockIdx.y;
    if (i >= n) return;
    int32_t d = pip_digit(sc, i, W, sbytes, w);
    if (d) {
        uint32_t pos = atomicAdd(&cursor[((size_t)w << W.cbk) + (uint32_t)((d < 0 ? -d : d) - 1)], 1u);
        sorted[(size_t)w * n + pos] = i | (d < 0 ? 0x80000000u : 0u);
    }
}

// The same counting sort with the window's counters in LDS (2^cbk <= 32768 words = 128 KB of the CU's 160 KB): grid (slices,
// windows), one 1024-thread workgroup per (slice of the points, window).  The global-atomic kernels above do one L2 atomic
// per (point, window) - 16.8 M of them at 2^20 points, 1.7 ms for histogram + scatter; here the atomics are ds_add(_rtn)
// and global memory sees each workgroup's counters once.
//   k_pip_hist_lds     counts of one slice            -> shist[window][slice][bucket]
//   k_pip_slice_scan   lane per bucket: exclusive prefix over the slices in place, total -> hist (then k_pip_scan_block -> offs)
//   k_pip_scatter_lds  cursors = offs + slice prefix in LDS; position = ds_add_rtn
constexpr uint32_t PIP_SORT_THREADS = 1024, PIP_SORT_MAX_CBK = 15, PIP_SLICES = 32;
__global__ void __launch_bounds__(PIP_SORT_THREADS) k_pip_hist_lds(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, pip_win W, uint32_t w0, uint32_t per,
                                                                   uint32_t* __restrict__ shist) {
    __shared__ uint32_t h[1u << PIP_SORT_MAX_CBK];
    const uint32_t s = blockIdx.x, w = w0 + blockIdx.y, nb = 1u << W.cbk;
    for (uint32_t b = threadIdx.x; b < nb; b += PIP_SORT_THREADS) h[b] = 0;
    __syncthreads();
    const uint32_t lo = s * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PIP_SORT_THREADS) {
        int32_t d = pip_digit(sc, i, W, sbytes, w);
        if (d) atomicAdd(&h[(uint32_t)((d < 0 ? -d : d) - 1)], 1u);
    }
    __syncthreads();
    uint32_t* out = shist + (((size_t)w * gridDim.x + s) << W.cbk);
    for (uint32_t b = threadIdx.x; b < nb; b += PIP_SORT_THREADS) out[b] = h[b];
}
__global__ void __launch_bounds__(WAVE) k_pip_slice_scan(uint32_t* __restrict__ shist, uint32_t nslice, uint32_t cbk, uint32_t g0, uint32_t gcount,
                                                         uint32_t* __restrict__ hist) {
    uint32_t t = blockIdx.x * WAVE + threadIdx.x;
    if (t >= gcount) return;
    uint32_t g = g0 + t, w = g >> cbk, b = g & ((1u << cbk) - 1u), run = 0;
    for (uint32_t s = 0; s < nslice; s++) {
        uint32_t* p = shist + ((((size_t)w * nslice + s) << cbk) | b);
        uint32_t v = *p;
        *p = run;
        run += v;
    }
    hist[g] = run;
}
// exclusive scan of a window's bucket counts, one 1024-thread workgroup per window: thread t owns 2^cbk / 1024 consecutive buckets
__global__ void __launch_bounds__(PIP_SORT_THREADS) k_pip_scan_block(const uint32_t* __restrict__ hist, uint32_t cbk, uint32_t* __restrict__ offs) {
    __shared__ uint32_t wsum[PIP_SORT_THREADS / WAVE];
    const uint32_t nb = 1u << cbk, per = nb / PIP_SORT_THREADS, t = threadIdx.x, lane = t & (WAVE - 1), wv = t / WAVE;
    const uint32_t* h = hist + ((size_t)blockIdx.x << cbk) + (size_t)t * per;
    uint32_t* o = offs + ((size_t)blockIdx.x << cbk) + (size_t)t * per;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) sum += h[j];
    uint32_t incl = sum;
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t v = __shfl_up(incl, d, WAVE);
        if ((int)lane >= d) incl += v;
    }
    if (lane == WAVE - 1) wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t k = 0; k < wv; k++) run += wsum[k];
    for (uint32_t j = 0; j < per; j++) {
        o[j] = run;
        run += h[j];
    }
}
__global__ void __launch_bounds__(PIP_SORT_THREADS) k_pip_scatter_lds(const uint8_t* __restrict__ sc, uint32_t sbytes, uint32_t n, pip_win W, uint32_t w0, uint32_t per,
                                                                      const uint32_t* __restrict__ shist, const uint32_t* __restrict__ offs,
                                                                      uint32_t* __restrict__ sorted) {
    __shared__ uint32_t h[1u << PIP_SORT_MAX_CBK];
    const uint32_t s = blockIdx.x, w = w0 + blockIdx.y, nb = 1u << W.cbk;
    const uint32_t* pre = shist + (((size_t)w * gridDim.x + s) << W.cbk);
    for (uint32_t b = threadIdx.x; b < nb; b += PIP_SORT_THREADS) h[b] = offs[((size_t)w << W.cbk) | b] + pre[b];
    __syncthreads();
    const uint32_t lo = s * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PIP_SORT_THREADS) {
        int32_t d = pip_digit(sc, i, W, sbytes, w);
        if (d) {
            uint32_t pos = atomicAdd(&h[(uint32_t)((d < 0 ? -d : d) - 1)], 1u);
            sorted[(size_t)w * n + pos] = i | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// buckets ordered by point count (descending) with a counting sort on min(count, 255)
// Each wave bins MSM_ORD_PER buckets per lane into an LDS histogram first, so the 256 global bins see
// one atomic per (wave, bin) instead of one per bucket.
constexpr uint32_t MSM_ORD_PER = 16;
__global__ void __launch_bounds__(WAVE) k_msm_order_hist(const uint32_t* __restrict__ hist, uint32_t total, uint32_t* __restrict__ chist) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += WAVE) h[i] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * WAVE * MSM_ORD_PER;
    for (uint32_t j = 0; j < MSM_ORD_PER; j++) {
        uint32_t t = base + j * WAVE + threadIdx.x;
        if (t < total) {
            uint32_t cc = hist[t] > 255 ? 255 : hist[t];
            atomicAdd(&h[255 - cc], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += WAVE)
        if (h[i]) atomicAdd(&chist[i], h[i]);
}
__global__ void k_msm_order_scan(uint32_t* __restrict__ chist) {     // 256 bins, one lane
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0;
    for (int i = 0; i < 256; i++) { uint32_t v = chist[i]; chist[i] = run; run += v; }
}
__global__ void __launch_bounds__(WAVE) k_msm_order_scatter(const uint32_t* __restrict__ hist, uint32_t total, uint32_t* __restrict__ chist, uint32_t* __restrict__ order) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += WAVE) h[i] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * WAVE * MSM_ORD_PER;
    uint32_t rank[MSM_ORD_PER], bin[MSM_ORD_PER];
#pragma unroll
    for (uint32_t j = 0; j < MSM_ORD_PER; j++) {
        uint32_t t = base + j * WAVE + threadIdx.x;
        bin[j] = 0xffffffffu;
        if (t < total) {
            uint32_t cc = hist[t] > 255 ? 255 : hist[t];
            bin[j] = 255 - cc;
            rank[j] = atomicAdd(&h[bin[j]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += WAVE)
        if (h[i]) h[i] = atomicAdd(&chist[i], h[i]);       // h[i] <- global base of this wave's run in bin i
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < MSM_ORD_PER; j++)
        if (bin[j] != 0xffffffffu) order[h[bin[j]] + rank[j]] = base + j * WAVE + threadIdx.x;
}

// field-generic memory helpers
__device__ __forceinline__ g1_aff ld_aff_blst(const uint32_t* w, const g1_aff*) { return ld_g1a_blst(w); }
__device__ __forceinline__ g2_aff ld_aff_blst(const uint32_t* w, const g2_aff*) { return ld_g2a_blst(w); }
__device__ __forceinline__ void st_aff_int(uint32_t* o, const g1_aff& q) { st_fp_int(o, q.x); st_fp_int(o + FPW, q.y); }
__device__ __forceinline__ void st_aff_int(uint32_t* o, const g2_aff& q) {
    st_fp_int(o, q.x.c0); st_fp_int(o + FPW, q.x.c1); st_fp_int(o + 2 * FPW, q.y.c0); st_fp_int(o + 3 * FPW, q.y.c1);
}
__device__ __forceinline__ g1_aff ld_aff_int(const uint32_t* w, const g1_aff*) { return g1_aff{ld_fp_int(w), ld_fp_int(w + FPW)}; }
__device__ __forceinline__ g2_aff ld_aff_int(const uint32_t* w, const g2_aff*) {
    return g2_aff{fp2{ld_fp_int(w), ld_fp_int(w + FPW)}, fp2{ld_fp_int(w + 2 * FPW), ld_fp_int(w + 3 * FPW)}};
}
__device__ __forceinline__ g1_jac soa_ld_jac(const uint4* b, size_t stride, size_t i, const g1_jac*) { return soa_ld_g1(b, stride, i); }
__device__ __forceinline__ g2_jac soa_ld_jac(const uint4* b, size_t stride, size_t i, const g2_jac*) { return soa_ld_g2(b, stride, i); }
__device__ __forceinline__ void soa_st_jac(uint4* b, size_t stride, size_t i, const g1_jac& a) { soa_st_g1(b, stride, i, a); }
__device__ __forceinline__ void soa_st_jac(uint4* b, size_t stride, size_t i, const g2_jac& a) { soa_st_g2(b, stride, i, a); }
__device__ __forceinline__ g1_jac ld_jac_int(const uint32_t* w, const g1_jac*) { return ld_g1_int(w); }
__device__ __forceinline__ g2_jac ld_jac_int(const uint32_t* w, const g2_jac*) { return ld_g2_int(w); }
__device__ __forceinline__ void st_jac_int(uint32_t* w, const g1_jac& a) { st_g1_int(w, a); }
__device__ __forceinline__ void st_jac_int(uint32_t* w, const g2_jac& a) { st_g2_int(w, a); }
__device__ __forceinline__ void st_jac_blst(uint32_t* w, const g1_jac& a) { st_g1_blst(w, a); }
__device__ __forceinline__ void st_jac_blst(uint32_t* w, const g2_jac& a) { st_g2_blst(w, a); }
template <class F> struct fld;                      // words of one coordinate in the internal AoS form
template <> struct fld<fp> { static constexpr int W = FPW; static constexpr int AFFB = 96; };
template <> struct fld<fp2> { static constexpr int W = 2 * FPW; static constexpr int AFFB = 192; };
// additions inside loops: inlined for G1 (operands stay in registers), the shared out-of-line body for G2 (code size)
__device__ __forceinline__ g1_jac padd(const g1_jac& a, const g1_jac& b) { return jac_add_body(a, b); }
__device__ __forceinline__ g2_jac padd(const g2_jac& a, const g2_jac& b) { return jac_add(a, b); }

// points converted once from the blst image to the device representation (2 multiplications per coordinate instead of per
// bucket addition): internal AoS
template <class F>
__global__ void __launch_bounds__(WAVE) k_pip_convert(const uint8_t* __restrict__ pts, uint32_t n, uint32_t* __restrict__ pts_int) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    aff<F> q = ld_aff_blst(reinterpret_cast<const uint32_t*>(pts + (size_t)i * fld<F>::AFFB), (const aff<F>*)nullptr);
    uint32_t* o = pts_int + (size_t)i * 2 * fld<F>::W;
    st_aff_int(o, q);
    o[FP_N] = aff_is_inf(q) ? 1u : 0u;                  // first pad word of x: "this point is the point at infinity" (tested once here, not per bucket addition)
}
// lane per (window, bucket); `order` lists the buckets so that a wave's lanes have similar counts
template <class F>
__global__ void __launch_bounds__(WAVE, sizeof(F) == sizeof(fp) ? 2 : 1) k_pip_bucket(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs,
                                                     const uint32_t* __restrict__ hist, const uint32_t* __restrict__ order, uint32_t n, uint32_t cbk,
                                                     uint32_t total, uint32_t g0, uint32_t gcount, uint4* __restrict__ buckets) {
    // buckets g0 .. g0 + gcount - 1 (a group of windows); order[g0 + t] lists them by load, as indices relative to g0
    uint32_t t = blockIdx.x * WAVE + threadIdx.x;
    if (t >= gcount) return;
    uint32_t g = g0 + (order ? order[g0 + t] : t);
    uint32_t w = g >> cbk, cnt = hist[g], off = offs[g];
    const uint32_t* srt = sorted + (size_t)w * n + off;
    xyzz<F> acc = xyzz_inf<F>();                        // extended Jacobian: 8M + 2S per mixed addition (curve.hpp)
    // the first point of a bucket initialises the accumulator (no addition: 3 % of all additions at 32 points per bucket); the
    // "operand is the point at infinity" test was made once by k_pip_convert
    if (cnt) {
        uint32_t e = srt[0];
        const uint32_t* pw = pts + (size_t)(e & 0x7fffffffu) * (2 * fld<F>::W);
        aff<F> q = ld_aff_int(pw, (const aff<F>*)nullptr);
        if (e >> 31) q.y = f_neg(q.y);
        F one_or_zero = f_select(pw[FP_N] != 0, f_zero<F>(), f_one<F>());
        acc = xyzz<F>{q.x, q.y, one_or_zero, one_or_zero};
    }
    for (uint32_t j = 1; j < cnt; j++) {
        uint32_t e = srt[j];
        const uint32_t* pw = pts + (size_t)(e & 0x7fffffffu) * (2 * fld<F>::W);
        aff<F> q = ld_aff_int(pw, (const aff<F>*)nullptr);
        if (e >> 31) q.y = f_neg(q.y);
        acc = xyzz_add_aff_flag(acc, q, pw[FP_N] != 0);
    }
    soa_st_jac(buckets, total, g, jac_from_xyzz(acc));
}
// lane per (window, segment of L buckets): sum_{j < L} (b0 + j + 1) * B_{b0 + j}
template <class F>
__global__ void __launch_bounds__(WAVE) k_pip_segred(const uint4* __restrict__ buckets, uint32_t total, uint32_t cbk, uint32_t L, uint32_t nseg_total,
                                                     uint32_t t0, uint32_t tcount, uint4* __restrict__ segout) {
    uint32_t t = blockIdx.x * WAVE + threadIdx.x;
    if (t >= tcount) return;
    t += t0;
    uint32_t segs_per_win = (1u << cbk) / L;
    uint32_t w = t / segs_per_win, b0 = (t % segs_per_win) * L;
    jac<F> S = jac_inf<F>(), T = jac_inf<F>();
    for (uint32_t j = L; j-- > 0;) {
        jac<F> B = soa_ld_jac(buckets, total, ((size_t)w << cbk) | (b0 + j), (const jac<F>*)nullptr);
        S = padd(S, B);               // running sums
        T = padd(T, S);               // T = sum (j + 1) * B_{b0 + j}
    }
    jac<F> acc = jac_inf<F>();        // [b0] S, b0 < 2^cbk
#pragma clang loop unroll(disable)
    for (int i = (int)cbk - 1; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((b0 >> i) & 1) acc = padd(acc, S);
    }
    acc = padd(acc, T);
    soa_st_jac(segout, nseg_total, t, acc);
}
// G1 lane teams for the latency-bound reductions: T = 2 or 4 ADJACENT lanes hold the same values; product k of a round runs in lane k mod T
// (round k / T) and the results are shared with DPP quad permutes (no LDS, no ds_bpermute).  A Jacobian addition costs 5 (T = 4) or 8
// (T = 2) multiplication times instead of 16, a doubling 3 or 5 instead of 7.  The formulas are jac_add_team / jac_dbl_team (curve.hpp).
template <int T>
struct team_quad_fp {
    uint32_t h;                       // threadIdx.x & (T - 1)
    // value of team lane J, in every lane of the team
    template <int J>
    __device__ __forceinline__ fp from(const fp& x) const {
        if (T == 4) return fp_quad_perm<J * 0x55>(x);
        return fp_quad_perm<(J ? 0xf5 : 0xa0)>(x);                  // [0,0,2,2] / [1,1,3,3]
    }
    __device__ __forceinline__ fp pick4(const fp& a0, const fp& a1, const fp& a2, const fp& a3) const {
        return fp_select((h & 1) != 0, fp_select((h & 2) != 0, a3, a1), fp_select((h & 2) != 0, a2, a0));
    }
    __device__ __forceinline__ void mul4(fp& r0, fp& r1, fp& r2, fp& r3, const fp& a0, const fp& b0, const fp& a1, const fp& b1, const fp& a2, const fp& b2, const fp& a3, const fp& b3) const {
        if (T == 4) {
            fp x = fp_mul(pick4(a0, a1, a2, a3), pick4(b0, b1, b2, b3));
            r0 = from<0>(x); r1 = from<1>(x); r2 = from<2>(x); r3 = from<3>(x);
        } else {
            const bool o = h != 0;
            fp x = fp_mul(fp_select(o, a1, a0), fp_select(o, b1, b0));
            fp y = fp_mul(fp_select(o, a3, a2), fp_select(o, b3, b2));
            r0 = from<0>(x); r1 = from<1>(x); r2 = from<0>(y); r3 = from<1>(y);
        }
    }
    __device__ __forceinline__ void mul3(fp& r0, fp& r1, fp& r2, const fp& a0, const fp& b0, const fp& a1, const fp& b1, const fp& a2, const fp& b2) const {
        fp r3;
        mul4(r0, r1, r2, r3, a0, b0, a1, b1, a2, b2, a2, b2);
    }
    __device__ __forceinline__ void sqr3(fp& r0, fp& r1, fp& r2, const fp& a0, const fp& a1, const fp& a2) const {
        if (T == 4) {
            fp x = fp_sqr(pick4(a0, a1, a2, a2));
            r0 = from<0>(x); r1 = from<1>(x); r2 = from<2>(x);
        } else {
            const bool o = h != 0;
            fp x = fp_sqr(fp_select(o, a1, a0));
            fp y = fp_sqr(a2);
            r0 = from<0>(x); r1 = from<1>(x); r2 = y;
        }
    }
    __device__ __forceinline__ void mul2(fp& r0, fp& r1, const fp& a0, const fp& b0, const fp& a1, const fp& b1) const {
        const bool o = (h & 1) != 0;
        fp x = fp_mul(fp_select(o, a1, a0), fp_select(o, b1, b0));
        if (T == 4) { r0 = fp_quad_perm<0xa0>(x); r1 = fp_quad_perm<0xf5>(x); }
        else { r0 = from<0>(x); r1 = from<1>(x); }
    }
    __device__ __forceinline__ void sqr2(fp& r0, fp& r1, const fp& a0, const fp& a1) const {
        const bool o = (h & 1) != 0;
        fp x = fp_sqr(fp_select(o, a1, a0));
        if (T == 4) { r0 = fp_quad_perm<0xa0>(x); r1 = fp_quad_perm<0xf5>(x); }
        else { r0 = from<0>(x); r1 = from<1>(x); }
    }
    __device__ __forceinline__ fp mul1(const fp& a, const fp& b) const { return fp_mul(a, b); }
};
// k_pip_segred for G1 with T lanes per segment (T = 2, 4): same sums, same order of additions, the products of every addition and doubling
// spread over the team.  tcount segments -> tcount * T lanes.
template <int T>
__global__ void __launch_bounds__(WAVE) k_pip_segred_team(const uint4* __restrict__ buckets, uint32_t total, uint32_t cbk, uint32_t L, uint32_t nseg_total,
                                                          uint32_t t0, uint32_t tcount, uint4* __restrict__ segout) {
    uint32_t lane = blockIdx.x * WAVE + threadIdx.x, t = lane / T;
    if (t >= tcount) return;            // whole teams leave together (T divides the wave)
    t += t0;
    const team_quad_fp<T> team{threadIdx.x & (T - 1)};
    uint32_t segs_per_win = (1u << cbk) / L;
    uint32_t w = t / segs_per_win, b0 = (t % segs_per_win) * L;
    g1_jac S = jac_inf<fp>(), Tt = jac_inf<fp>();
#pragma clang loop unroll(disable)
    for (uint32_t j = L; j-- > 0;) {
        g1_jac B = soa_ld_g1(buckets, total, ((size_t)w << cbk) | (b0 + j));
        S = jac_add_team(S, B, team);
        Tt = jac_add_team(Tt, S, team);
    }
    g1_jac acc = jac_inf<fp>();
#pragma clang loop unroll(disable)
    for (int i = (int)cbk - 1; i >= 0; i--) {
        acc = jac_dbl_team(acc, team);
        if ((b0 >> i) & 1) acc = jac_add_team(acc, S, team);
    }
    acc = jac_add_team(acc, Tt, team);
    if ((threadIdx.x & (T - 1)) == 0) soa_st_g1(segout, nseg_total, t, acc);
}
// grid (windows, nsplit): partial sums of a window's segment values
template <class F>
__global__ void __launch_bounds__(WAVE) k_pip_winpart(const uint4* __restrict__ segout, uint32_t nseg_total, uint32_t segs_per_win, uint32_t w0, uint32_t* __restrict__ part) {
    uint32_t w = w0 + blockIdx.x, sp = blockIdx.y, nsplit = gridDim.y;
    jac<F> acc = jac_inf<F>();
    for (uint32_t j = sp * WAVE + threadIdx.x; j < segs_per_win; j += WAVE * nsplit)
        acc = padd(acc, soa_ld_jac(segout, nseg_total, (size_t)w * segs_per_win + j, (const jac<F>*)nullptr));
    for (int d = 32; d >= 1; d >>= 1) {
        jac<F> o = shfl_down_struct(acc, d);
        acc = padd(acc, o);
    }
    if (threadIdx.x == 0) st_jac_int(part + ((size_t)w * nsplit + sp) * (3 * fld<F>::W), acc);
}
// Lane-parallel G1 doubling for the serial doubling chains: every lane holds the same point; the three independent
// products of each of the first two rounds of dbl-2009-l run in lanes 0, 1, 2 of ONE multiplier call and are then
// broadcast (3 multiplication times per doubling instead of 7).  Same carry/reduce pattern as jac_dbl.
__device__ __forceinline__ fp fp_bcast(const fp& a, int src) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __shfl(a.l[i], src, WAVE);
    return r;
}
struct team_wave_fp {               // every lane of the wave holds the same point; lanes 0, 1, 2 take the products
    __device__ __forceinline__ void mul3(fp& r0, fp& r1, fp& r2, const fp& a0, const fp& b0, const fp& a1, const fp& b1, const fp& a2, const fp& b2) const {
        const uint32_t l = threadIdx.x;
        fp r = fp_mul(fp_select(l == 0, a0, fp_select(l == 1, a1, a2)), fp_select(l == 0, b0, fp_select(l == 1, b1, b2)));
        r0 = fp_bcast(r, 0); r1 = fp_bcast(r, 1); r2 = fp_bcast(r, 2);
    }
    __device__ __forceinline__ void sqr3(fp& r0, fp& r1, fp& r2, const fp& a0, const fp& a1, const fp& a2) const {
        const uint32_t l = threadIdx.x;
        fp r = fp_sqr(fp_select(l == 0, a0, fp_select(l == 1, a1, a2)));
        r0 = fp_bcast(r, 0); r1 = fp_bcast(r, 1); r2 = fp_bcast(r, 2);
    }
    __device__ __forceinline__ fp mul1(const fp& a, const fp& b) const { return fp_mul(a, b); }
};
__device__ __forceinline__ g1_jac g1_dbl_coop(const g1_jac& p) { return jac_dbl_team(p, team_wave_fp{}); }
__device__ __forceinline__ g1_jac dbl_coop(const g1_jac& p) { return g1_dbl_coop(p); }
__device__ __forceinline__ g2_jac dbl_coop(const g2_jac& p) { return g2_dbl_coop(p, threadIdx.x & ~7u, threadIdx.x & 7u); }
__device__ __forceinline__ g1_jac bcast0(const g1_jac& a) { return g1_jac{fp_bcast(a.x, 0), fp_bcast(a.y, 0), fp_bcast(a.z, 0)}; }
__device__ __forceinline__ g2_jac bcast0(const g2_jac& a) {
    return g2_jac{fp2{fp_bcast(a.x.c0, 0), fp_bcast(a.x.c1, 0)}, fp2{fp_bcast(a.y.c0, 0), fp_bcast(a.y.c1, 0)}, fp2{fp_bcast(a.z.c0, 0), fp_bcast(a.z.c1, 0)}};
}
// one wave per window: R_w = sum of its nsplit partial sums, then 2^(off_w) * R_w
template <class F>
__global__ void __launch_bounds__(WAVE) k_pip_winsum(const uint32_t* __restrict__ part, uint32_t nsplit, pip_win W, uint32_t w0, uint32_t* __restrict__ winout) {
    uint32_t w = w0 + blockIdx.x;
    jac<F> acc = jac_inf<F>();
    for (uint32_t j = threadIdx.x; j < nsplit; j += WAVE) acc = padd(acc, ld_jac_int(part + ((size_t)w * nsplit + j) * (3 * fld<F>::W), (const jac<F>*)nullptr));
    int top = 32;
    while (top >= 1 && (uint32_t)top >= nsplit) top >>= 1;          // lanes >= nsplit hold the neutral element
    for (int d = top; d >= 1; d >>= 1) {
        jac<F> o = shfl_down_struct(acc, d);
        acc = padd(acc, o);
    }
    acc = bcast0(acc);
    uint32_t sh = pip_off(W, w);
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < sh; i++) acc = dbl_coop(acc);
    if (threadIdx.x == 0) st_jac_int(winout + (size_t)w * (3 * fld<F>::W), acc);
}
// one wave: sum of the window values -> blst image (Jacobian)
template <class F>
__global__ void __launch_bounds__(WAVE) k_pip_final(const uint32_t* __restrict__ winout, uint32_t nparts, uint32_t* __restrict__ out) {
    jac<F> acc = jac_inf<F>();
    for (uint32_t j = threadIdx.x; j < nparts; j += WAVE) acc = padd(acc, ld_jac_int(winout + (size_t)j * (3 * fld<F>::W), (const jac<F>*)nullptr));
    int top = 32;
    while (top >= 1 && (uint32_t)top >= nparts) top >>= 1;
    for (int d = top; d >= 1; d >>= 1) {
        jac<F> o = shfl_down_struct(acc, d);
        acc = padd(acc, o);
    }
    if (threadIdx.x == 0) st_jac_blst(out, acc);
}

// sum of k Jacobian blst images `stride` words apart -> blst image: the merge of the per-device partials of a point-sharded MSM
__device__ __forceinline__ g1_jac ld_jac_blst(const uint32_t* w, const g1_jac*) { return ld_g1_blst(w); }
__device__ __forceinline__ g2_jac ld_jac_blst(const uint32_t* w, const g2_jac*) { return ld_g2_blst(w); }
template <class F>
__global__ void __launch_bounds__(WAVE) k_jac_sum_blst(const uint32_t* __restrict__ parts, uint32_t k, uint32_t stride, uint32_t* __restrict__ out) {
    jac<F> acc = jac_inf<F>();
    for (uint32_t j = threadIdx.x; j < k; j += WAVE) acc = padd(acc, ld_jac_blst(parts + (size_t)j * stride, (const jac<F>*)nullptr));
    int top = 32;
    while (top >= 1 && (uint32_t)top >= k) top >>= 1;
    for (int d = top; d >= 1; d >>= 1) {
        jac<F> o = shfl_down_struct(acc, d);
        acc = padd(acc, o);
    }
    if (threadIdx.x == 0) st_jac_blst(out, acc);
}

// ------------------------------------------------------------------------------------------
// Signature side of large batches: bucket fold + bilinearity instead of n 64-bit scalar multiplications.
//   e(-G1, sum_i [r_i]S_i) = prod_{w,d} e(-[d 2^(cw)]G1, B_{w,d}),   B_{w,d} = sum of the S_i whose w-th c-bit
//   digit of r_i is d  (r_i = sum_w d_{i,w} 2^(cw)).
// The 64/c x 2^c bucket sums B (counting sort by digit, then mixed additions only: 64/c per tuple instead of
// 64 doublings + ~32 additions) become 64/c x 2^c EXTRA MILLER PAIRS against constant G1 points, so no
// bucket reduction, window doubling chain or other serial tail is needed at all.  The verdict and the
// final-exponentiated GT value are unchanged; sum [r_i]S_i itself (BLST's AggrSign, fetch_stage 3) is folded
// from the buckets only on demand (k_sig_fold).
// ------------------------------------------------------------------------------------------
// signatures converted once from the blst image to the device representation: internal AoS, 4 x FPW words
__global__ void __launch_bounds__(WAVE) k_sig_convert(const uint8_t* __restrict__ sets, uint32_t n, uint32_t* __restrict__ pts_int) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    g2_aff q = ld_g2a_blst(reinterpret_cast<const uint32_t*>(sets + (size_t)i * 320 + 128));
    uint32_t* o = pts_int + (size_t)i * 4 * FPW;
    st_fp_int(o, q.x.c0); st_fp_int(o + FPW, q.x.c1); st_fp_int(o + 2 * FPW, q.y.c0); st_fp_int(o + 3 * FPW, q.y.c1);
}
// slot g = (w << c) | d  ->  -[d 2^(cw)]G1 (Jacobian, internal AoS); d = 0 gives infinity (pair skipped)
__global__ void __launch_bounds__(WAVE) k_sig_consts(uint32_t c, uint32_t total, uint32_t* __restrict__ out) {
    uint32_t g = blockIdx.x * WAVE + threadIdx.x;
    if (g >= total) return;
    uint64_t sc = (uint64_t)(g & ((1u << c) - 1u)) << (c * (g >> c));
    g1_aff ng{fp_from_const(k::G1_X), fp_from_const(k::G1_NEG_Y)};
    st_g1_int(out + (size_t)g * G1W, jac_mul_u64(ng, sc));
}
// L lanes (a power of two <= 64) per bucket slot: lane s adds entries s, s+L, ... of the bucket's sorted list,
// then the L partial sums are folded with wave shuffles.  Result -> Miller pair n + g = (consts[g], B_g).
__global__ void __launch_bounds__(WAVE) k_sig_bucket(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs,
                                                     const uint32_t* __restrict__ hist, uint32_t n, uint32_t c, uint32_t lshift, uint32_t total,
                                                     const uint32_t* __restrict__ consts, uint4* __restrict__ H, uint4* __restrict__ P, size_t stride, size_t base) {
    uint32_t t = blockIdx.x * WAVE + threadIdx.x;
    uint32_t L = 1u << lshift, g = t >> lshift, s = t & (L - 1);
    xyzz<fp2> part = xyzz_inf<fp2>();                   // extended Jacobian: 8M + 2S per mixed addition (curve.hpp)
    if (g < total) {
        uint32_t w = g >> c, cnt = hist[g], off = offs[g];
        const uint32_t* srt = sorted + (size_t)w * n + off;
        for (uint32_t j = s; j < cnt; j += L) {
            const uint32_t* pw = pts + (size_t)srt[j] * (4 * FPW);
            g2_aff q{fp2{ld_fp_int(pw), ld_fp_int(pw + FPW)}, fp2{ld_fp_int(pw + 2 * FPW), ld_fp_int(pw + 3 * FPW)}};
            part = xyzz_add_aff(part, q);
        }
    }
    g2_jac acc = jac_from_xyzz(part);
    for (uint32_t d = L >> 1; d >= 1; d >>= 1) {
        g2_jac o = shfl_down_struct(acc, (int)d);
        acc = jac_add(acc, o);
    }
    if (g < total && s == 0) {
        soa_st_g2(H, stride, base + g, acc);
        soa_st_g1(P, stride, base + g, ld_g1_int(consts + (size_t)g * G1W));
    }
}
// On demand (fetch_stage 3): sum [r_i]S_i = sum_w 2^(cw) sum_d d B_{w,d}; lane w folds window w with running
// sums, then a Horner pass over the windows.  One wave, slow, never on the verification path.
__global__ void __launch_bounds__(WAVE) k_sig_fold(const uint4* __restrict__ H, size_t stride, uint32_t n, uint32_t nwin, uint32_t c, uint32_t* __restrict__ agg_out) {
    __shared__ uint32_t wsum[32 * G2W];
    uint32_t w = threadIdx.x;
    if (w < nwin) {
        g2_jac S = jac_inf<fp2>(), T = jac_inf<fp2>();
        for (uint32_t d = (1u << c) - 1; d >= 1; d--) {
            S = jac_add(S, soa_ld_g2(H, stride, (size_t)n + ((w << c) | d)));
            T = jac_add(T, S);
        }
        st_g2_int(wsum + (size_t)w * G2W, T);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g2_jac acc = ld_g2_int(wsum + (size_t)(nwin - 1) * G2W);
        for (uint32_t j = nwin - 1; j-- > 0;) {
            for (uint32_t i = 0; i < c; i++) acc = jac_dbl(acc);
            acc = jac_add(acc, ld_g2_int(wsum + (size_t)j * G2W));
        }
        st_g2_blst(agg_out, acc);
    }
}

// ------------------------------------------------------------------------------------------
// k_deser: one lane per tuple: compressed public key (48 B) + signature (96 B) -> validated
// SignatureSet record (320 B, BLST images) + status byte.  Replaces PublicKey.fromBytes /
// Signature.fromBytes per tuple (bls_sig_io.nim:42-58,81-99).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) k_deser(const uint8_t* __restrict__ pks, const uint8_t* __restrict__ msgs, const uint8_t* __restrict__ sigs,
                                                uint32_t n, uint32_t dflags, uint8_t* __restrict__ sets, uint8_t* __restrict__ status,
                                                uint32_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    g1_aff pk;
    g2_aff sg;
    const size_t pkb = (dflags & DESER_F_PK_UNCOMPRESSED) ? 96 : 48, sgb = (dflags & DESER_F_SIG_UNCOMPRESSED) ? 192 : 96;
    uint8_t st = deserialize_tuple(pk, sg, pks + (size_t)i * pkb, sigs + (size_t)i * sgb, dflags);
    status[i] = st;
    if (st != DESER_OK) atomicOr(flags + 2, 1u);
    uint32_t* o = reinterpret_cast<uint32_t*>(sets + (size_t)i * 320);
    bool ok = st == DESER_OK;
    pk = g1_aff{fp_select(ok, pk.x, fp_zero()), fp_select(ok, pk.y, fp_zero())};
    sg = g2_aff{fp2_select(ok, sg.x, fp2_zero()), fp2_select(ok, sg.y, fp2_zero())};
    st_fp_blst(o, pk.x); st_fp_blst(o + 12, pk.y);
    const uint8_t* m = msgs + (size_t)i * 32;
    for (int j = 0; j < 8; j++) o[24 + j] = (uint32_t)m[4 * j] | ((uint32_t)m[4 * j + 1] << 8) | ((uint32_t)m[4 * j + 2] << 16) | ((uint32_t)m[4 * j + 3] << 24);
    st_fp_blst(o + 32, sg.x.c0); st_fp_blst(o + 44, sg.x.c1); st_fp_blst(o + 56, sg.y.c0); st_fp_blst(o + 68, sg.y.c1);
}

// ------------------------------------------------------------------------------------------
// MultiSignatureSet.combine (blst_min_pubkey_sig_core.nim:570-647): same-message pre-aggregation
//   s_i: chain seeded with rnd ITSELF, u64 words 3,2,1,0 of every digest, zeros skipped (:588-606)
//   pk' = sum [s_i]PK_i, sig' = sum [s_i]S_i  (the reference's two 64-bit Pippenger calls, :629-646)
// ------------------------------------------------------------------------------------------
// `finish` (to affine, core :172-177 / blst_p{1,2}_to_affine): Jacobian blst images -> affine blst images
__global__ void k_finish_affine(const uint32_t* __restrict__ p1, const uint32_t* __restrict__ p2, uint32_t* __restrict__ out_pk, uint32_t* __restrict__ out_sig) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    g1_jac a = ld_g1_blst(p1);
    if (jac_is_inf(a)) {
        for (int i = 0; i < 24; i++) out_pk[i] = 0;
    } else {
        fp zi = fp_inv(a.z), zi2 = fp_sqr(zi);
        st_fp_blst(out_pk, fp_mul(a.x, zi2));
        st_fp_blst(out_pk + 12, fp_mul(a.y, fp_mul(zi2, zi)));
    }
    g2_jac b{fp2{ld_fp_blst(p2), ld_fp_blst(p2 + 12)}, fp2{ld_fp_blst(p2 + 24), ld_fp_blst(p2 + 36)}, fp2{ld_fp_blst(p2 + 48), ld_fp_blst(p2 + 60)}};
    if (jac_is_inf(b)) {
        for (int i = 0; i < 48; i++) out_sig[i] = 0;
    } else {
        fp2 zi = fp2_inv(b.z), zi2 = fp2_sqr(zi);
        fp2 x = fp2_mul(b.x, zi2), y = fp2_mul(b.y, fp2_mul(zi2, zi));
        st_fp_blst(out_sig, x.c0); st_fp_blst(out_sig + 12, x.c1); st_fp_blst(out_sig + 24, y.c0); st_fp_blst(out_sig + 36, y.c1);
    }
}

// ------------------------------------------------------------------------------------------
// aggregateVerify (bls_sig_min_pubkey.nim:153-199 -> ContextCoreAggregateVerify, core :305-414):
// e(G1, sig) == prod e(pk_i, H(m_i)), distinct messages of any length, no blinding.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WAVE) k_hash_var(const uint8_t* __restrict__ msgs, const uint32_t* __restrict__ offs, uint32_t n, dst_t dst,
                                                   uint4* __restrict__ H, size_t stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_jac h = hash_to_g2(msgs + offs[i], offs[i + 1] - offs[i], dst.b, dst.len);
    soa_st_g2(H, stride, i, h);
}
// 32-byte messages, packed -> offset 96 of 320-byte records (the layout k_hash_map reads); the other bytes are not read
__global__ void __launch_bounds__(WAVE) k_aggv_records(const uint8_t* __restrict__ msgs, uint32_t n, uint8_t* __restrict__ recs) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < 32; j++) recs[(size_t)i * 320 + 96 + j] = msgs[(size_t)i * 32 + j];
}
// pairs 0..n-1: P = pk_i (affine, Z = 1); pair n: (P, Q) = (-G1, sig)
__global__ void __launch_bounds__(WAVE) k_aggv_setup(const uint8_t* __restrict__ pks, uint32_t n, int with_sig, const uint32_t* __restrict__ sig, uint4* __restrict__ H,
                                                     uint4* __restrict__ P, size_t stride, uint32_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        g1_aff pk = ld_g1a_blst(reinterpret_cast<const uint32_t*>(pks + (size_t)i * 96));
        if (aff_is_inf(pk)) atomicOr(flags, 1u);                // BLST_PK_IS_INFINITY -> update() false
        soa_st_g1(P, stride, i, jac_from_aff(pk));
    } else if (i == n && with_sig) {
        soa_st_g1(P, stride, n, g1_jac{fp_from_const(k::G1_X), fp_from_const(k::G1_NEG_Y), fp_one()});
        soa_st_g2(H, stride, n, jac_from_aff(ld_g2a_blst(sig)));
    }
}

// ------------------------------------------------------------------------------------------
// Batch signer / input generator: per tuple publicFromSecret (core :118-133:
// sk == 0 or sk >= r -> false; pk = affine([sk]G1)) and coreSign (core :230-251: sig = affine([sk]H(msg))),
// written as a SignatureSet record.  Variable-time scalar multiplication: test/bench input generation only.
// ------------------------------------------------------------------------------------------
BLS_HD bool sk_load_check(uint32_t (&kk)[8], const uint8_t* sk) {
    uint32_t any = 0, borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        kk[j] = (uint32_t)sk[4 * j] | ((uint32_t)sk[4 * j + 1] << 8) | ((uint32_t)sk[4 * j + 2] << 16) | ((uint32_t)sk[4 * j + 3] << 24);
        any |= kk[j];
        uint64_t d = (uint64_t)kk[j] - k::R_ORDER[j] - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return any != 0 && borrow != 0;                                  // 0 < sk < r
}
__global__ void __launch_bounds__(WAVE) k_sign_pk(const uint8_t* __restrict__ sks, const uint8_t* __restrict__ msgs, uint32_t n, uint8_t* __restrict__ sets,
                                                  uint8_t* __restrict__ status, uint32_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    uint32_t kk[8];
    bool ok = sk_load_check(kk, sks + (size_t)i * 32);
    status[i] = ok ? 0 : 1;
    if (!ok) atomicOr(flags + 2, 1u);
    uint32_t* o = reinterpret_cast<uint32_t*>(sets + (size_t)i * 320);
    g1_aff g{fp_from_const(k::G1_X), fp_from_const(k::G1_Y)};
    g1_jac a = jac_mul_256(g, kk);
    fp zi = fp_inv(a.z), zi2 = fp_sqr(zi);                           // z != 0 for 0 < sk < r
    fp x = fp_select(ok, fp_mul(a.x, zi2), fp_zero()), y = fp_select(ok, fp_mul(a.y, fp_mul(zi2, zi)), fp_zero());
    st_fp_blst(o, x); st_fp_blst(o + 12, y);
    const uint8_t* m = msgs + (size_t)i * 32;
    for (int j = 0; j < 8; j++) o[24 + j] = (uint32_t)m[4 * j] | ((uint32_t)m[4 * j + 1] << 8) | ((uint32_t)m[4 * j + 2] << 16) | ((uint32_t)m[4 * j + 3] << 24);
}
__global__ void __launch_bounds__(WAVE) k_sign_sig(const uint8_t* __restrict__ sks, const uint8_t* __restrict__ msgs, uint32_t n, dst_t dst,
                                                   uint8_t* __restrict__ sets) {
    uint32_t i = blockIdx.x * WAVE + threadIdx.x;
    if (i >= n) return;
    uint32_t kk[8];
    bool ok = sk_load_check(kk, sks + (size_t)i * 32);
    uint8_t msg[32];
    for (int j = 0; j < 32; j++) msg[j] = msgs[(size_t)i * 32 + j];
    g2_jac h = hash_to_g2(msg, 32, dst.b, dst.len);
    g2_jac a = jac_mul_256_jac(h, kk);
    fp2 zi = fp2_inv(a.z), zi2 = fp2_sqr(zi);
    fp2 x = fp2_select(ok, fp2_mul(a.x, zi2), fp2_zero()), y = fp2_select(ok, fp2_mul(a.y, fp2_mul(zi2, zi)), fp2_zero());
    uint32_t* o = reinterpret_cast<uint32_t*>(sets + (size_t)i * 320) + 32;
    st_fp_blst(o, x.c0); st_fp_blst(o + 12, x.c1); st_fp_blst(o + 24, y.c0); st_fp_blst(o + 36, y.c1);
}

// shard state for the device-resident exchange: 576-byte committed state, then the ok word (1 = no update failed), zero padding
__global__ void k_pack_blob(const uint32_t* __restrict__ states, const uint32_t* __restrict__ flags, uint32_t* __restrict__ blob) {
    for (uint32_t i = threadIdx.x; i < MI355_BLS_BLOB_BYTES / 4; i += blockDim.x) blob[i] = i < 144 ? states[i] : (i == 144 ? (flags[0] == 0 ? 1u : 0u) : 0u);
}

// Jacobian SoA -> AoS copies for stage inspection
__global__ void k_export_g2(const uint4* __restrict__ H, size_t stride, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_jac a = soa_ld_g2(H, stride, i);
    st_g2_blst(out + (size_t)i * 72, a);
}
__global__ void k_export_g1(const uint4* __restrict__ P, size_t stride, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_jac a = soa_ld_g1(P, stride, i);
    st_g1_blst(out + (size_t)i * 36, a);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Context
// ------------------------------------------------------------------------------------------
struct msm_ws {
    size_t cap_n = 0;                 // capacity of d_pts in bytes
    uint32_t cap_total = 0;
    uint8_t* d_pts = nullptr;
    uint8_t* d_sc = nullptr;
    uint32_t* pts_int = nullptr;
    uint32_t *hist = nullptr, *offs = nullptr, *cursor = nullptr, *sorted = nullptr, *order = nullptr, *chist = nullptr, *winout = nullptr, *out = nullptr, *part = nullptr, *shist = nullptr;
    hipEvent_t ev_fork = nullptr, ev_bucketed = nullptr, ev_join = nullptr, ev_g[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t gs3 = nullptr;        // a third window group's stream (MI355_BLS_MSM_CUTS experiments; the context's main and side streams carry the first two)
    uint4 *buckets = nullptr, *segout = nullptr;
};
static void msm_free(msm_ws* m) {
    void* b[] = {m->d_pts, m->d_sc, m->pts_int, m->hist, m->offs, m->cursor, m->sorted, m->order, m->chist, m->winout, m->out, m->buckets, m->segout, m->part, m->shist};
    for (void* x : b)
        if (x) (void)hipFree(x);
    hipEvent_t ev[] = {m->ev_fork, m->ev_bucketed, m->ev_join, m->ev_g[0], m->ev_g[1], m->ev_g[2], m->ev_g[3]};
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    if (m->gs3) (void)hipStreamDestroy(m->gs3);
    *m = msm_ws();
}

struct mi355_bls_ctx {
    int device = 0;
    size_t cap = 0;          // sets the pipeline workspace holds at once (larger batches are sliced)
    size_t cap_io = 0;       // sets the staging buffers d_sets / d_comp / d_status / d_r hold (>= cap; grown on demand, io_reserve)
    size_t stride = 0;       // pairs capacity (cap + 1 rounded up to 64)
    uint32_t num_threads = 4096;
    uint32_t nblk_cap = 64;
    // device buffers
    uint8_t* d_sets = nullptr;       // staging for host-pointer calls
    uint8_t* d_rnd = nullptr;
    uint64_t* d_r = nullptr;
    uint4* d_H = nullptr;
    uint4* d_M = nullptr;            // the two mapped points per message before cofactor clearing (2 x cap Jacobian slots)
    size_t mstride = 0;
    uint4* d_P = nullptr;
    uint4* d_lines = nullptr;
    // bucket fold of the signatures (batches of >= SIG_BUCKET_MIN tuples)
    uint32_t* d_sig_pts = nullptr;   // signatures in the device representation, cap x 4 x FPW words
    uint32_t* d_sig_sorted = nullptr;// counting sort by digit: nwin x cap tuple indices
    uint32_t* d_sig_hist = nullptr;  // 3 x SIG_SLOTS_MAX: histogram, offsets, cursors
    uint32_t* d_sig_consts = nullptr;// -[d 2^(cw)]G1 for the window widths c = 4 and c = 8
    uint32_t sig_c = 0, sig_slots = 0; // window width / bucket slots of the last batch (0: per-tuple multiplications)
    bool agg_valid = false;
    uint32_t* d_agg = nullptr;
    uint32_t* d_agg1 = nullptr;      // G1 aggregate (blst_p1 image)
    uint8_t* d_msg = nullptr;        // message (<= 4096 B) + signature staging
    uint8_t* d_comp = nullptr;       // compressed wire-format staging: cap x (48 + 32 + 96) bytes
    uint8_t* d_status = nullptr;     // per-tuple deserialisation status
    hipEvent_t ev_deser0 = nullptr, ev_deser1 = nullptr;
    float deser_ms = 0.f;
    uint32_t* d_lpart = nullptr;
    uint32_t* d_L = nullptr;
    uint32_t* d_states = nullptr;    // up to 64 committed states (slot 0 = own)
    uint32_t* d_gt = nullptr;
    uint32_t* d_blob = nullptr;      // shard state + ok word for the device-resident exchange (MI355_BLS_BLOB_BYTES)
    uint32_t* d_blob_out = nullptr;  // where shard submits write the blob: d_blob, or a caller's device buffer (set_shard_blob_device)
    bool fv_pending = false;         // a finalverify_blobs submit has not been waited for
    hipStream_t fv_stream = nullptr;
    uint32_t* d_flags = nullptr;     // [0] = update-failed flag, [1] = verdict, [2] = some tuple failed to deserialise / sign, [3] = verdict of finalverify_blobs
                                     // (a word and a GT buffer of its own: a blob merge may be in flight on another stream while this context takes the next shard)
    uint32_t* d_gt_fv = nullptr;     // GT of the last finalverify_blobs
    bool gt_is_fv = false;           // fetch_stage(4): the last GT came from finalverify_blobs
    uint32_t* d_carry = nullptr;     // 2 x 8 seed words: blinding-chain state of the chunk that a slice boundary cuts (capacity-free batches)
    bool fail_next_enqueue = false;  // test hook (mi355_bls_debug_fail_next_enqueue)
    uint32_t* h_flags = nullptr;     // pinned host copy of d_flags[0..1] (asynchronous submit / wait)
    bool pending = false;            // a submitted batch has not been waited for yet
    bool coop = true;                // small batches: lane-cooperative kernels (latency) instead of one lane per item (throughput)
    bool wide_recorded = false;      // ev_lp (end of the whole-chip kernels) has been recorded at least once
    hipStream_t pending_stream = nullptr;
    hipStream_t side = nullptr;      // fork / join stream of latency-mode calls (independent stages beside each other)
    uint32_t* d_export = nullptr;
    hipEvent_t ev[9] = {};
    // A batch larger than the workspace runs in slices; the slices of ONE call are pipelined over up to three workspaces - this
    // context's and two internal ones (lanes), created at the first sliced call, each on a stream of its own - like the batches of
    // three callers (run_shard).
    mi355_bls_ctx* lane[2] = {nullptr, nullptr};
    hipStream_t lane_st[2] = {nullptr, nullptr};
    hipEvent_t lane_ev[2] = {nullptr, nullptr};
    hipEvent_t ev_sl0 = nullptr, ev_blind[3] = {nullptr, nullptr, nullptr};
    bool is_lane = false;
    hipEvent_t ev_hm = nullptr, ev_lp = nullptr;   // inside the hash stage (after k_hash_map) and the line-product stage (after k_lineprod)
    hipEvent_t ev_s0 = nullptr, ev_l0 = nullptr;   // start of the signature side (on its stream) and of the tuple pairs' Miller lines: the stage timers of forked calls
    float ktimes[4] = {};         // k_hash_map, k_hash_clear, k_lineprod, k_lineprod2 of the last batch call
    uint32_t slots = 1024;        // wave slots at one wave per SIMD: 4 x CUs
    size_t last_n = 0;
    bool have_gt = false;
    float timings[8] = {};
    dst_t dst;
    xmd32_consts xmd;                // message-independent SHA-256 words of expand_message_xmd for this DST
    std::vector<uint64_t> h_r;       // host-computed scalar chains (serial blinding chain, combine)
    msm_ws* msm = nullptr;           // lazily sized MSM workspace
    std::vector<uint8_t> av_pks, av_msgs;      // streaming aggregateVerify (mi355_bls_aggv_*): the pairs collected so far
    std::vector<uint32_t> av_offs;
    bool av_failed = false;
    msm_ws* msm2 = nullptr;          // a second one: combine runs its G1 and its G2 Pippenger side by side
};

constexpr uint32_t SIG_SLOTS_MAX = 2048;     // 8 windows x 256 digits
constexpr size_t SIG_WIDE_MIN = 40000;       // from here 8-bit digits (2048 extra pairs, 8 additions per tuple) beat 4-bit ones (256, 15)

// HIP spreads streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run strictly in turn: a host
// with many SMALL batches in flight (one context + stream each) gets 1.4 M verifications/s with 4 queues and 2.3 M/s with 8
// (tests/gpu_probe_small.py; more than 8 abort in the runtime).  The variable is read when the HIP runtime initialises.  The library
// does NOT touch the process environment on its own (round 3 did, from a load-time constructor: setenv is not thread-safe, it changed
// queue behaviour for every HIP user of the process, and it silently did nothing when HIP was already up): the host sets
// GPU_MAX_HW_QUEUES=8 itself, or calls this ONCE, from its main thread, before anything initialises HIP.
extern "C" int mi355_bls_recommend_hw_queues(void) {
    const char* no = getenv("MI355_BLS_NO_ENV");
    if (no && no[0] == '1') return 0;                             // the host forbids environment edits
    return setenv("GPU_MAX_HW_QUEUES", "8", 0) == 0 ? 1 : 0;       // 0 = overwrite flag: a value the host has set stays
}

static const char DST_SIG[] = "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";   // bls_sig_min_pubkey.nim:31

extern "C" const char* mi355_bls_last_error(void) { return g_err.c_str(); }

// How this library was built (build.sh passes the three macros to the host compile): "aligned=1 dpp_combine=off stamp=<sha256 of the
// sources>".  aligned=0 means the instruction-alignment post-pass (tools/align_isa.py) was skipped - an 8-byte VALU stream then issues
// ~23 % slower - which build.sh only does when BLS_NO_ALIGN=1 asks for it; bench.py prints the string with its numbers.
#ifndef BLS_BUILD_ALIGNED
#define BLS_BUILD_ALIGNED -1
#endif
#ifndef BLS_BUILD_STAMP
#define BLS_BUILD_STAMP "unknown"
#endif
#define BLS_STR2(x) #x
#define BLS_STR(x) BLS_STR2(x)
extern "C" const char* mi355_bls_build_info(void) { return "aligned=" BLS_STR(BLS_BUILD_ALIGNED) " dpp_combine=off stamp=" BLS_BUILD_STAMP; }

extern "C" void mi355_bls_ctx_destroy(mi355_bls_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (int k = 0; k < 2; k++) {
        if (c->lane_st[k]) {
            (void)hipStreamSynchronize(c->lane_st[k]);
            (void)hipStreamDestroy(c->lane_st[k]);
        }
        if (c->lane_ev[k]) (void)hipEventDestroy(c->lane_ev[k]);
        if (c->lane[k]) mi355_bls_ctx_destroy(c->lane[k]);
    }
    if (c->ev_sl0) (void)hipEventDestroy(c->ev_sl0);
    for (auto& e : c->ev_blind)
        if (e) (void)hipEventDestroy(e);
    void* bufs[] = {c->d_sets, c->d_rnd, c->d_r, c->d_H, c->d_M, c->d_P, c->d_lines, c->d_sig_pts, c->d_sig_sorted, c->d_sig_hist, c->d_sig_consts, c->d_agg, c->d_agg1, c->d_msg, c->d_comp, c->d_status, c->d_lpart, c->d_L, c->d_states, c->d_gt, c->d_gt_fv, c->d_carry, c->d_blob, c->d_flags, c->d_export};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_hm) (void)hipEventDestroy(c->ev_hm);
    if (c->ev_lp) (void)hipEventDestroy(c->ev_lp);
    if (c->ev_s0) (void)hipEventDestroy(c->ev_s0);
    if (c->ev_l0) (void)hipEventDestroy(c->ev_l0);
    if (c->ev_deser0) (void)hipEventDestroy(c->ev_deser0);
    if (c->ev_deser1) (void)hipEventDestroy(c->ev_deser1);
    if (c->msm2) {
        msm_free(c->msm2);
        delete c->msm2;
    }
    if (c->msm) {
        msm_free(c->msm);
        delete c->msm;
    }
    delete c;
}

// everything of ctx_create that can fail after the context object exists: any failure destroys it (no leaked device buffers)
static int ctx_build(mi355_bls_ctx* c, int device, size_t max_sets) {
    c->device = device;
    c->msm = new msm_ws();
    c->msm2 = new msm_ws();
    c->cap = max_sets;
    c->cap_io = max_sets;
    c->stride = ((max_sets + 1 + SIG_SLOTS_MAX + 63) / 64) * 64;          // tuple pairs + the extra pair(s) of the signature side
    std::memset(&c->dst, 0, sizeof(c->dst));
    c->dst.len = sizeof(DST_SIG) - 1;
    std::memcpy(c->dst.b, DST_SIG, c->dst.len);
    c->xmd = xmd32_precompute(c->dst.b, c->dst.len);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->slots = 4u * (uint32_t)prop.multiProcessorCount;
    size_t nwaves = c->stride / 64;
    // k_lineprod hands over (68 steps x nblk ranges x 64 lanes) partial products; nblk <= slots / 68, and never more ranges than waves of pairs
    uint32_t nblk = c->slots / N_LINES;
    if (nblk < 1) nblk = 1;
    if (nblk > 64) nblk = 64;
    if (nblk > nwaves) nblk = (uint32_t)nwaves;
    c->nblk_cap = nblk;
#define ALLOC(p, bytes) HIPCHK(hipMalloc((void**)&(p), (bytes)))
    ALLOC(c->d_sets, max_sets * 320);
    ALLOC(c->d_rnd, 32);
    ALLOC(c->d_r, c->stride * 8);
    ALLOC(c->d_H, c->stride * 6 * 64);
    c->mstride = ((2 * max_sets + 63) / 64) * 64;
    ALLOC(c->d_M, c->mstride * 6 * 64);
    ALLOC(c->d_P, c->stride * 3 * 64);
    ALLOC(c->d_lines, c->stride * 6 * 64 * (size_t)N_LINES);
    ALLOC(c->d_sig_pts, max_sets * 4 * FPW * 4);
    ALLOC(c->d_sig_sorted, max_sets * 16 * 4);
    ALLOC(c->d_sig_hist, 3 * SIG_SLOTS_MAX * 4);
    ALLOC(c->d_sig_consts, 2 * SIG_SLOTS_MAX * G1W * 4);
    ALLOC(c->d_agg, 288);
    ALLOC(c->d_agg1, 144);
    ALLOC(c->d_msg, 4096 + 192 + 64 + 288);      // message | affine signature | pad | Jacobian signature (AggregateSignature overloads)
    ALLOC(c->d_comp, max_sets * 320);          // wire-format staging: keys (<= 96 B) | messages (32 B) | signatures (<= 192 B)
    ALLOC(c->d_status, max_sets);
    ALLOC(c->d_lpart, (size_t)N_LINES * (c->nblk_cap * (WAVE + 1) + 64) * F12W * 4);     // per-lane partial products of k_lineprod (+ k_fold's first-level results)
    ALLOC(c->d_L, (size_t)N_LINES * F12W * 4);
    ALLOC(c->d_states, 64 * 576);
    ALLOC(c->d_gt, 576);
    ALLOC(c->d_gt_fv, 576);
    ALLOC(c->d_carry, 64);
    ALLOC(c->d_blob, MI355_BLS_BLOB_BYTES);
    c->d_blob_out = c->d_blob;
    ALLOC(c->d_flags, 16);
    ALLOC(c->d_export, c->stride * 288 + 2048 * 2 * G1W * 4);
#undef ALLOC
    HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK(hipHostMalloc((void**)&c->h_flags, 1024, hipHostMallocDefault));    // words 0..3 flags, 4..11 staging copy of rnd, 16..159 shard state
    for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreate(&c->ev_hm));
    HIPCHK(hipEventCreate(&c->ev_lp));
    HIPCHK(hipEventCreate(&c->ev_s0));
    HIPCHK(hipEventCreate(&c->ev_l0));
    HIPCHK(hipEventCreate(&c->ev_deser0));
    HIPCHK(hipEventCreate(&c->ev_deser1));
    k_sig_consts<<<(256 + WAVE - 1) / WAVE, WAVE>>>(4, 256, c->d_sig_consts);
    k_sig_consts<<<(2048 + WAVE - 1) / WAVE, WAVE>>>(8, 2048, c->d_sig_consts + (size_t)SIG_SLOTS_MAX * G1W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

extern "C" int mi355_bls_ctx_create(mi355_bls_ctx** out, int device, size_t max_sets) {
    if (!out || max_sets == 0 || max_sets > (1u << 30)) return MI355_BLS_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        g_err = "no such HIP device";
        return MI355_BLS_ERR_HIP;
    }
    HIPCHK(hipSetDevice(device));
    auto* c = new mi355_bls_ctx();
    int rc = ctx_build(c, device, max_sets);
    if (rc) {
        std::string keep = g_err;
        mi355_bls_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return 0;
}

// The staging buffers of the entry points around the batch path (wire-format arrays, keys of fastAggregateVerify, the records
// fromBytes writes, combine's inputs and scalars) grow on demand: like the reference's procs, which take any openArray, no entry
// point refuses an input for its size.  (The pipeline workspace itself stays at max_sets: larger batches are sliced, run_shard.)
static int io_reserve(mi355_bls_ctx* c, size_t n) {
    if (n <= c->cap_io) return 0;
    if (c->pending) {
        g_err = "a batch submitted on this context has not been waited for";
        return MI355_BLS_ERR_ARG;
    }
    size_t want = n + n / 4;
    HIPCHK(hipSetDevice(c->device));
    // the new buffers first: if one allocation fails the context keeps its old buffers and capacity (entry points that take
    // device-resident input never come through here and would otherwise launch on null pointers)
    void* nb[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[4] = {want * 320, want * 320, want, (want > c->stride ? want : c->stride) * 8};
    for (int i = 0; i < 4; i++) {
        hipError_t e = hipMalloc(&nb[i], bytes[i]);
        if (e != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipFree(nb[j]);
            g_err = std::string("io_reserve: hipMalloc: ") + hipGetErrorString(e);
            return MI355_BLS_ERR_HIP;
        }
    }
    // only this context's own work can still read the old buffers, and no call is pending (checked above; blocking calls return after
    // their stream has drained): the fork stream and the stream of the last call are waited for, never the whole device (other
    // contexts keep running)
    // (pending_stream is cleared by every wait: a handle kept from an earlier call may belong to a stream the host has destroyed since)
    hipError_t se = c->side ? hipStreamSynchronize(c->side) : hipSuccess;
    if (se == hipSuccess && c->pending_stream) se = hipStreamSynchronize(c->pending_stream);
    if (se != hipSuccess) {
        for (int i = 0; i < 4; i++) (void)hipFree(nb[i]);
        g_err = std::string("io_reserve: hipStreamSynchronize: ") + hipGetErrorString(se);
        return MI355_BLS_ERR_HIP;
    }
    void** bufs[] = {(void**)&c->d_sets, (void**)&c->d_comp, (void**)&c->d_status, (void**)&c->d_r};
    for (int i = 0; i < 4; i++) {
        if (*bufs[i]) (void)hipFree(*bufs[i]);
        *bufs[i] = nb[i];
    }
    c->cap_io = want;
    return 0;
}

extern "C" int mi355_bls_ctx_set_num_threads(mi355_bls_ctx* c, uint32_t nt) {
    if (!c || nt == 0) return MI355_BLS_ERR_ARG;
    c->num_threads = nt;
    return 0;
}

extern "C" int mi355_bls_ctx_set_cooperative(mi355_bls_ctx* c, int on) {
    if (!c) return MI355_BLS_ERR_ARG;
    c->coop = on != 0;
    return 0;
}

extern "C" void mi355_bls_chunk_range(size_t n_total, uint32_t num_threads, uint32_t lo, uint32_t hi, size_t* first, size_t* count) {
    size_t B = n_total < num_threads ? n_total : num_threads;
    if (B == 0 || lo >= hi) {
        *first = 0;
        *count = 0;
        return;
    }
    if (hi > B) hi = (uint32_t)B;
    size_t base = n_total / B, rem = n_total % B;
    auto off = [&](size_t c) { return c < rem ? (base + 1) * c : base * c + rem; };
    *first = off(lo);
    *count = off(hi) - off(lo);      // off(B) == n_total
}

// The two inherently serial SHA-256 chains of the reference run on the HOST (one GPU lane needs ~4 us per compression:
// 65 536 links = a quarter of a second; one CPU core does them in a few milliseconds):
//   batchVerifySerial's single blinding chain (core :502-505, :545-556): seed = SHA256(rnd), then per tuple
//   seed <- SHA256(seed) until the low u64 is non-zero;
//   combine's chain (core :588-606): seeded with rnd itself, u64 words 3,2,1,0 of every digest, zeros skipped.
static void host_sha256_32(const uint32_t (&in)[8], uint32_t (&out)[8]) {        // SHA-256 of 32 bytes given as 8 big-endian words
    uint32_t w[16] = {in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], 0x80000000u, 0, 0, 0, 0, 0, 0, 256};
    sha256_init(out);
    sha256_compress_core(out, w);
}
static void host_serial_chain(const uint8_t rnd[32], size_t n, uint64_t* out) {
    uint32_t seed[8], nx[8];
    for (int i = 0; i < 8; i++) nx[i] = ((uint32_t)rnd[4 * i] << 24) | ((uint32_t)rnd[4 * i + 1] << 16) | ((uint32_t)rnd[4 * i + 2] << 8) | rnd[4 * i + 3];
    host_sha256_32(nx, seed);
    for (size_t j = 0; j < n; j++) {
        uint64_t r;
        do {
            host_sha256_32(seed, nx);
            for (int i = 0; i < 8; i++) seed[i] = nx[i];
            r = digest_low_u64_le(seed);
        } while (r == 0);
        out[j] = r;
    }
}
static void host_combine_chain(const uint8_t rnd[32], size_t n, uint64_t* out) {
    uint32_t seed[8], nx[8];
    for (int i = 0; i < 8; i++) seed[i] = ((uint32_t)rnd[4 * i] << 24) | ((uint32_t)rnd[4 * i + 1] << 16) | ((uint32_t)rnd[4 * i + 2] << 8) | rnd[4 * i + 3];
    int avail = 0;
    for (size_t i = 0; i < n; i++) {
        for (;;) {
            if (avail == 0) {
                host_sha256_32(seed, nx);
                for (int j = 0; j < 8; j++) seed[j] = nx[j];
                avail = 4;
            }
            avail--;
            uint64_t w = (uint64_t)bswap32(seed[2 * avail]) | ((uint64_t)bswap32(seed[2 * avail + 1]) << 32);   // LE u64 word `avail`
            if (w != 0) {
                out[i] = w;
                break;
            }
        }
    }
}

// Miller lines of pairs 0 .. npairs-1 (the last `extra` of them are the bucket pairs of the signature side): the
// 8-lanes-per-pair kernel while that does not take more waves than the chip has slots.  Whole-chip batches in latency mode
// (coop): the tuple pairs fill the chip exactly, so the few extra pairs would be a second round of waves that takes as long as
// the first (2.2 ms at 3 % occupancy); with 8 lanes per pair they take ~1 ms instead.  In throughput mode (several batches
// in flight) that second round overlaps other batches' kernels and one lane per pair is the cheaper form.
static inline int tail_threads(const mi355_bls_ctx* c) { return c->coop ? TAIL_THREADS : TAIL_THREADS_TP; }
static void launch_lines(mi355_bls_ctx* c, uint32_t npairs, uint32_t extra, hipStream_t st) {
    if (c->coop && (npairs + 3) / 4 <= c->slots / 2) {
        k_lines_coop<16><<<(npairs + 3) / 4, WAVE, 0, st>>>(c->d_P, c->d_H, 0, npairs, c->stride, c->d_lines);
    } else if (c->coop && (npairs + 7) / 8 <= c->slots) {
        k_lines_coop<8><<<(npairs + 7) / 8, WAVE, 0, st>>>(c->d_P, c->d_H, 0, npairs, c->stride, c->d_lines);
    } else if (c->coop && extra && extra < npairs && (extra + 7) / 8 <= c->slots &&
               (npairs + WAVE - 1) / WAVE > c->slots * (((npairs - extra + WAVE - 1) / WAVE + c->slots - 1) / c->slots)) {
        // the extra pairs would start one more round of waves: 8 lanes each instead
        uint32_t main_pairs = npairs - extra;
        k_lines<<<(main_pairs + WAVE - 1) / WAVE, WAVE, 0, st>>>(c->d_P, c->d_H, 0, main_pairs, c->stride, c->d_lines);
        if ((extra + 3) / 4 <= c->slots / 2) k_lines_coop<16><<<(extra + 3) / 4, WAVE, 0, st>>>(c->d_P, c->d_H, main_pairs, extra, c->stride, c->d_lines);
        else k_lines_coop<8><<<(extra + 7) / 8, WAVE, 0, st>>>(c->d_P, c->d_H, main_pairs, extra, c->stride, c->d_lines);
    } else {
        k_lines<<<(npairs + WAVE - 1) / WAVE, WAVE, 0, st>>>(c->d_P, c->d_H, 0, npairs, c->stride, c->d_lines);
    }
}

// The per-step products of the Miller lines of pairs 0 .. npairs-1 -> d_L (68 step products).  mid_ev: recorded between the wide
// kernel and the fold of its partials.
static int enqueue_line_products(mi355_bls_ctx* c, uint32_t npairs, hipStream_t st, hipEvent_t mid_ev) {
    uint32_t nblk = c->slots / N_LINES;
    if (nblk < 1) nblk = 1;
    if (nblk > c->nblk_cap) nblk = c->nblk_cap;
    uint32_t m = (npairs + WAVE * nblk - 1) / (WAVE * nblk);
    if (m < 1) m = 1;
    nblk = (npairs + WAVE * m - 1) / (WAVE * m);
    // every lane hands its partial product over (64 x nblk per step).  Throughput mode: k_lineprod2's 68 waves fold them,
    // 15 sequential Fp12 products per lane + one shuffle tree (least total work); latency mode: k_fold on the lane-cooperative
    // engine, 64 per block and then the nblk block results (one caller, 65 536 tuples: 1.8 -> 0.35 ms)
    // 1: the assembly loop (32-bit byte offsets inside one step's 24 planes); 2: the compiled loop
    const int per_lane = (uint64_t)c->stride * 16 * 24 + (uint64_t)npairs * 16 < (1ull << 32) ? 1 : 2;
    k_lineprod<<<dim3(N_LINES, nblk), WAVE, 0, st>>>(c->d_lines, npairs, c->stride, m, c->d_lpart, nblk, per_lane);
    if (mid_ev) HIPCHK(hipEventRecord(mid_ev, st));
    if (c->coop) {
        size_t first_last = (size_t)(nblk - 1) * WAVE * m;             // lanes past the last pair hold 1: not folded
        uint32_t live = (nblk - 1) * WAVE + (npairs - first_last < WAVE ? (uint32_t)(npairs - first_last) : WAVE);
        uint32_t per = 1;                                              // two levels of about sqrt(live) dependent products each
        while (per * per < live) per++;
        uint32_t nb1 = (live + per - 1) / per;
        uint32_t* mid = c->d_lpart + (size_t)N_LINES * c->nblk_cap * WAVE * F12W;
        k_fold<<<dim3(N_LINES, nb1), TAIL_THREADS, 0, st>>>(c->d_lpart, nblk * WAVE, per, live - (nb1 - 1) * per, nb1 > 1 ? mid : c->d_L);
        if (nb1 > 1) k_fold<<<dim3(N_LINES, 1), TAIL_THREADS, 0, st>>>(mid, nb1, nb1, nb1, c->d_L);
    } else {
        k_lineprod2<<<N_LINES, WAVE, 0, st>>>(c->d_lpart, nblk * WAVE, c->d_L);
    }
    return 0;
}

// Enqueues everything up to the committed state (d_states slot 0) of ONE SLICE of a shard: tuples [tuple_base, tuple_base + n) of the
// global batch, n <= capacity, records at d_sets (device memory).  chunk_lo / chunk_cnt: the blinding chains that overlap the slice.
// The three producers of Miller pairs are independent until the lines: hashing (k_hash_map, k_hash_clear), [r]PK (k_pkmul) and
// the signature side (bucket fold).  A batch that fills the chip runs them one after the other on the caller's stream (each is a
// whole-chip kernel).  A small batch in latency mode runs the last two on the context's side stream beside the hashing: they
// are all latency-bound there (a few waves each), so this takes about a millisecond off the call.
static int run_pairs(mi355_bls_ctx* c, const uint8_t* d_sets, size_t n, hipStream_t st);
// c: the workspace this slice runs in (the caller's context or one of its lanes); p: the caller's context, which holds what the slices
// of one call share - the random bytes, the carried chain state, the host-computed serial chain.  blind_done (may be null) is
// recorded behind the blinding kernel: the next slice's chains continue from the state this one leaves.
static int run_slice(mi355_bls_ctx* c, mi355_bls_ctx* p, const uint8_t* d_sets, size_t n_total, uint32_t nchunks, uint32_t chunk_lo, uint32_t chunk_cnt,
                     size_t tuple_base, size_t n, int serial, size_t serial_off, uint32_t slice, hipStream_t st, hipEvent_t blind_done) {
    HIPCHK(hipEventRecord(c->ev[0], st));
    if (serial) {
        HIPCHK(hipMemcpyAsync(c->d_r, p->h_r.data() + serial_off, n * 8, hipMemcpyHostToDevice, st));
    } else {
        k_blind<<<(chunk_cnt + WAVE - 1) / WAVE, WAVE, 0, st>>>(p->d_rnd, n_total, nchunks, chunk_lo, chunk_cnt, tuple_base, n, p->d_carry + 8 * (slice & 1),
                                                                p->d_carry + 8 * ((slice + 1) & 1), c->d_r);
    }
    if (blind_done) HIPCHK(hipEventRecord(blind_done, st));
    return run_pairs(c, d_sets, n, st);
}
// Everything behind the blinding scalars (d_r[0 .. n) are ready on `st`): hashing, [r]PK, the signature side, Miller lines, line
// products, the committed state of these n tuples in d_states slot 0.
static int run_pairs(mi355_bls_ctx* c, const uint8_t* d_sets, size_t n, hipStream_t st) {
    uint32_t n32 = (uint32_t)n;
    uint32_t nb = (n32 + WAVE - 1) / WAVE;
    HIPCHK(hipEventRecord(c->ev[1], st));
    const bool fork = c->coop && c->side && n32 <= 16 * c->slots;       // pk + signature side beside the hashing
    // A whole-chip batch of ONE caller (latency mode): the signature side and the Miller lines of its extra pairs run on the side
    // stream beside the hashing.  The tuple pairs then fill the chip's wave slots exactly once; behind them the extra pairs
    // would be a second round of waves (or ~1 ms of the 8-lanes-per-pair kernel).  Throughput mode keeps everything on the
    // caller's stream: with several batches in flight the nearly empty second round overlaps other batches' kernels, and
    // folding the 2048 bucket sums further (to 64 per-bit sums, or to one sum per window) so that fewer extra pairs remain was
    // measured SLOWER per pipelined batch (+0.5 ms and +2.5 ms: the fold is a chain of small dependent kernels on the batch's
    // critical path, the 2048 extra pairs are 3 % more of two embarrassingly parallel kernels).
    const bool fork_sig = !fork && c->coop && c->side;
    hipStream_t sd = fork ? c->side : st;                               // [r]PK
    hipStream_t ss = (fork || fork_sig) ? c->side : st;                 // signature side
    if (fork || fork_sig) HIPCHK(hipStreamWaitEvent(c->side, c->ev[1], 0));
    // ---- hashing (caller's stream)
    k_hash_map<<<(2 * n32 + WAVE - 1) / WAVE, WAVE, 0, st>>>(d_sets, n32, c->dst, c->xmd, c->d_M, c->mstride);
    HIPCHK(hipEventRecord(c->ev_hm, st));
    if (c->coop && (n32 + 3) / 4 <= c->slots / 2)            // 16 lanes per message while that leaves half the wave slots free (at 4 096 messages it fills the chip and gains nothing)
        k_hash_clear_coop<16><<<(n32 + 3) / 4, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
    else if (c->coop && (n32 + 7) / 8 <= c->slots)
        k_hash_clear_coop<8><<<(n32 + 7) / 8, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
    else
        k_hash_clear<<<nb, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
    HIPCHK(hipEventRecord(c->ev[2], st));
    // ---- [r]PK
    k_pkmul<<<nb, WAVE, 0, sd>>>(d_sets, n32, c->d_r, c->d_P, c->stride, c->d_flags);
    HIPCHK(hipEventRecord(c->ev[3], sd));
    // ---- signature side as a bucket fold: sig_slots extra Miller pairs n .. n + sig_slots - 1 (every batch size: for a
    // handful of tuples the 256 nearly empty buckets are still cheaper than one 64-bit G2 multiplication per tuple, which is a
    // 3 ms chain of doublings when nothing hides its latency)
    uint32_t cw = n >= SIG_WIDE_MIN ? 8 : 4, nwin = 64 / cw, total = nwin << cw;
    {
        msm_win W{nwin, cw, 0};
        uint32_t *hist = c->d_sig_hist, *offs = hist + SIG_SLOTS_MAX, *cursor = offs + SIG_SLOTS_MAX;
        HIPCHK(hipEventRecord(c->ev_s0, ss));
        HIPCHK(hipMemsetAsync(hist, 0, (size_t)total * 4, ss));
        k_sig_convert<<<nb, WAVE, 0, ss>>>(d_sets, n32, c->d_sig_pts);
        k_msm_hist<<<dim3(nb, nwin), WAVE, 0, ss>>>(reinterpret_cast<const uint8_t*>(c->d_r), 8, n32, W, cw, hist);
        k_msm_scan<<<nwin, WAVE, 0, ss>>>(hist, cw, offs, cursor);
        k_msm_scatter<<<dim3(nb, nwin), WAVE, 0, ss>>>(reinterpret_cast<const uint8_t*>(c->d_r), 8, n32, W, cw, cursor, c->d_sig_sorted);
        uint32_t per = n32 >> cw, lshift = 0;                          // expected entries per bucket; ~16 per lane
        const uint32_t per_lane_min = c->coop ? 16u : 64u;      // throughput mode: fewer, longer lanes (the fold of a bucket's lanes is pure overhead)
        while (lshift < 6 && (per >> (lshift + 1)) >= per_lane_min) lshift++;
        // small batches leave most of the chip idle: more lanes per bucket (down to ~2 entries per lane) shorten the kernel
        while (lshift < 6 && ((total << (lshift + 1)) <= 16 * c->slots) && (per >> (lshift + 1)) >= 2) lshift++;
        k_sig_bucket<<<((total << lshift) + WAVE - 1) / WAVE, WAVE, 0, ss>>>(c->d_sig_pts, c->d_sig_sorted, offs, hist, n32, cw, lshift, total,
                                                                             c->d_sig_consts + (cw == 8 ? (size_t)SIG_SLOTS_MAX * G1W : 0), c->d_H, c->d_P,
                                                                             c->stride, (size_t)n32);
        c->sig_c = cw;
        c->sig_slots = total;
        c->agg_valid = false;
    }
    // ---- Miller lines and their products per step
    uint32_t npairs = n32 + total;
    if (fork_sig) {
        k_lines<<<(total + WAVE - 1) / WAVE, WAVE, 0, ss>>>(c->d_P, c->d_H, n32, total, c->stride, c->d_lines);
        HIPCHK(hipEventRecord(c->ev[4], ss));
        HIPCHK(hipEventRecord(c->ev_l0, st));
        launch_lines(c, n32, 0, st);
        HIPCHK(hipStreamWaitEvent(st, c->ev[4], 0));                    // join
    } else {
        HIPCHK(hipEventRecord(c->ev[4], ss));
        if (fork) HIPCHK(hipStreamWaitEvent(st, c->ev[4], 0));          // join
        HIPCHK(hipEventRecord(c->ev_l0, st));
        launch_lines(c, npairs, total, st);
    }
    HIPCHK(hipEventRecord(c->ev[5], st));
    {
        int rcp = enqueue_line_products(c, npairs, st, c->ev_lp);
        if (rcp) return rcp;
    }
    c->wide_recorded = true;
    HIPCHK(hipEventRecord(c->ev[6], st));
    k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, 1, c->d_gt, c->d_flags + 1, 144, 0);
    HIPCHK(hipEventRecord(c->ev[7], st));
    HIPCHK(hipGetLastError());
    c->last_n = n;
    c->have_gt = false;
    c->gt_is_fv = false;
    return 0;
}

// chunk of the parallel_chunks partition (parallel_chunks.nim:42-66) that tuple t of n_total falls into, B chunks
static inline uint32_t chunk_of_tuple(size_t n_total, uint32_t B, size_t t) {
    size_t base = n_total / B, rem = n_total % B, cut = (base + 1) * rem;
    return (uint32_t)(t < cut ? t / (base + 1) : rem + (t - cut) / base);
}

__global__ void k_or_flag(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && src[0]) atomicOr(dst, src[0]);
}
// the internal workspaces of pipelined slices: same device, same capacity, throughput mode (no fork streams: the slices overlap each other)
// Returns the number of lanes usable (0 .. want): a lane is published only when its workspace, its stream and its event all exist,
// and a lane that cannot be created (out of memory: a workspace is ~29 KB per set) is not an error - the slices then run on fewer
// workspaces, down to this context's own (the serial slice loop of round 3).  Negative: the events every sliced call needs failed.
static int ensure_lanes(mi355_bls_ctx* c, int want) {
    if (!c->ev_sl0) HIPCHK(hipEventCreateWithFlags(&c->ev_sl0, hipEventDisableTiming));
    for (auto& e : c->ev_blind)
        if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int have = 0;
    for (int k = 0; k < want && k < 2; k++) {
        if (c->lane[k]) {
            have = k + 1;
            continue;
        }
        mi355_bls_ctx* x = nullptr;
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        if (mi355_bls_ctx_create(&x, c->device, c->cap) != 0) x = nullptr;
        if (x && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
        if (x && s && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        if (!x || !s || !e) {
            if (e) (void)hipEventDestroy(e);
            if (s) (void)hipStreamDestroy(s);
            if (x) mi355_bls_ctx_destroy(x);
            (void)hipGetLastError();                // an out-of-memory error is sticky until read
            (void)hipSetDevice(c->device);
            break;
        }
        x->is_lane = true;
        x->coop = false;
        c->lane_st[k] = s;
        c->lane_ev[k] = e;
        c->lane[k] = x;
        have = k + 1;
    }
    return have;
}
// A shard = chunks [chunk_lo, chunk_lo + chunk_cnt) = tuples [tuple_base, tuple_base + n) of the global batch -> committed state in
// d_states slot 0.  The reference's cache holds per-thread pairing contexts only and accepts any input.len
// (bls_batch_verifier.nim:108-119,141); here the workspace is sized for `cap` tuples, so a larger shard is processed in
// ceil(n / cap) balanced SLICES on the same stream: every slice commits its own state (its own signature-side pairs folded in),
// the running product is kept in slot 1 (blst_pairing_merge, blst_abi.nim:508), the blinding chain of a chunk that a slice
// boundary cuts is carried over (k_blind).  src_dev: the shard's records in device memory, or src_host: in host memory
// (staged slice by slice through d_sets).  After a sliced call fetch_stage(0..3) shows the LAST slice.
static int run_shard(mi355_bls_ctx* c, const uint8_t* src_dev, const uint8_t* src_host, size_t n_total, uint32_t nchunks, uint32_t chunk_lo, uint32_t chunk_cnt,
                     size_t tuple_base, size_t n, int serial, const uint8_t rnd[32], hipStream_t st) {
    (void)chunk_lo; (void)chunk_cnt;
    HIPCHK(hipSetDevice(c->device));
    if (c->fail_next_enqueue) {                                 // test hook: an enqueue failure after earlier shards of a multi-device call went out
        c->fail_next_enqueue = false;
        g_err = "injected enqueue failure (mi355_bls_debug_fail_next_enqueue)";
        return MI355_BLS_ERR_HIP;
    }
    std::memcpy(c->h_flags + 4, rnd, 32);                      // pinned staging: the copy below is then truly asynchronous
    HIPCHK(hipMemcpyAsync(c->d_rnd, c->h_flags + 4, 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
    if (serial) {
        c->h_r.resize(n);
        host_serial_chain(rnd, n, c->h_r.data());
    }
    const size_t nslices = (n + c->cap - 1) / c->cap;
    if (nslices == 1) {
        const uint8_t* d = src_dev ? src_dev : c->d_sets;
        if (!src_dev) HIPCHK(hipMemcpyAsync(c->d_sets, src_host, n * 320, hipMemcpyHostToDevice, st));
        uint32_t c_lo = serial ? 0 : chunk_of_tuple(n_total, nchunks, tuple_base), c_hi = serial ? 1 : chunk_of_tuple(n_total, nchunks, tuple_base + n - 1) + 1;
        return run_slice(c, c, d, n_total, nchunks, c_lo, c_hi - c_lo, tuple_base, n, serial, 0, 0, st, nullptr);
    }
    // ---- several slices: pipelined over the workspaces of this context and of up to two lanes, each on its own stream.  Slice i starts
    // when slice i - 1 (on another workspace) has finished hashing and its public-key multiplications, exactly as bench.py staggers the
    // batches of three callers (`after`): the slices then sit at different stages and the serial tail of one runs beside the wide
    // kernels of another.  What orders them: the blinding chains (a chunk cut by a slice boundary continues from the state the previous
    // slice's k_blind left: ev_blind), and each workspace's own stream.  Every workspace keeps the running product of ITS slices in
    // slot 1 of its d_states; at the end the lanes' products are copied over and multiplied in (an Fp12 product commutes).  The last
    // slice always runs in this context's own workspace, so fetch_stage(0..3) shows it as before.
    int nl = nslices >= 3 ? 3 : 2;                            // workspaces used, this context's included
    {
        int have = ensure_lanes(c, nl - 1);
        if (have < 0) return have;
        nl = have + 1;                                         // fewer lanes than wanted (memory): fewer slices in flight, down to one after the other
    }
    HIPCHK(hipEventRecord(c->ev_sl0, st));                     // rnd uploaded, flags cleared
    for (int k = 0; k < nl - 1; k++) {
        mi355_bls_ctx* x = c->lane[k];
        x->num_threads = c->num_threads;
        x->dst = c->dst;
        x->xmd = c->xmd;
        HIPCHK(hipStreamWaitEvent(c->lane_st[k], c->ev_sl0, 0));
        HIPCHK(hipMemsetAsync(x->d_flags, 0, 12, c->lane_st[k]));
    }
    bool used[3] = {false, false, false};
    mi355_bls_ctx* prev = nullptr;
    size_t done = 0;
    for (uint32_t slice = 0; done < n; slice++) {
        size_t left = nslices - slice, cnt = (n - done + left - 1) / left;          // balanced: never a sliver at the end
        size_t t0 = tuple_base + done;
        uint32_t c_lo = serial ? 0 : chunk_of_tuple(n_total, nchunks, t0), c_hi = serial ? 1 : chunk_of_tuple(n_total, nchunks, t0 + cnt - 1) + 1;
        const int L = (int)((nslices - 1 - slice) % (size_t)nl);                    // the last slice on this context's own workspace
        mi355_bls_ctx* x = L ? c->lane[L - 1] : c;
        hipStream_t sx = L ? c->lane_st[L - 1] : st;
        if (slice) {
            if (!serial) HIPCHK(hipStreamWaitEvent(sx, c->ev_blind[(slice - 1) % 3], 0));       // the chain state this slice continues from
            if (prev != x) HIPCHK(hipStreamWaitEvent(sx, prev->ev[3], 0));                      // stagger: behind the previous slice's hashing and [r]PK
        }
        const uint8_t* d = src_dev ? src_dev + 320 * done : x->d_sets;
        if (!src_dev) HIPCHK(hipMemcpyAsync(x->d_sets, src_host + 320 * done, cnt * 320, hipMemcpyHostToDevice, sx));
        int rc = run_slice(x, c, d, n_total, nchunks, c_lo, c_hi - c_lo, t0, cnt, serial, done, slice, sx, c->ev_blind[slice % 3]);
        if (rc) {
            // earlier slices are still running on the lane streams and read the caller's records and this context's chain state: drain
            // them before the error goes back (the caller may free its buffers then); the error of the failed enqueue is what is reported
            std::string keep = g_err;
            for (int k = 0; k < nl - 1; k++) (void)hipStreamSynchronize(c->lane_st[k]);
            (void)hipStreamSynchronize(st);
            g_err = keep;
            return rc;
        }
        k_state_mul<<<1, TAIL_THREADS, 0, sx>>>(x->d_states, 1, used[L] ? 1 : 0, used[L] ? 0 : -1);
        used[L] = true;
        prev = x;
        done += cnt;
    }
    for (int k = 0; k < nl - 1; k++) {
        if (!used[k + 1]) continue;
        mi355_bls_ctx* x = c->lane[k];
        HIPCHK(hipEventRecord(c->lane_ev[k], c->lane_st[k]));
        HIPCHK(hipStreamWaitEvent(st, c->lane_ev[k], 0));
        HIPCHK(hipMemcpyAsync(c->d_states + (size_t)(2 + k) * 144, x->d_states + 144, 576, hipMemcpyDeviceToDevice, st));
        k_or_flag<<<1, 1, 0, st>>>(c->d_flags, x->d_flags);                         // an infinity public key in a lane's slice fails the call
        k_state_mul<<<1, TAIL_THREADS, 0, st>>>(c->d_states, 1, 1, 2 + k);
    }
    k_state_mul<<<1, TAIL_THREADS, 0, st>>>(c->d_states, 0, 1, -1);
    HIPCHK(hipGetLastError());
    return 0;
}

static int collect_timings(mi355_bls_ctx* c, int last_ev) {
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    for (int i = 0; i < 4; i++) c->ktimes[i] = 0;
    if (last_ev == 7) {                           // batch path: per-kernel split of the two-kernel stages
        HIPCHK(hipEventElapsedTime(&c->ktimes[0], c->ev[1], c->ev_hm));
        HIPCHK(hipEventElapsedTime(&c->ktimes[1], c->ev_hm, c->ev[2]));
        HIPCHK(hipEventElapsedTime(&c->ktimes[2], c->ev[5], c->ev_lp));
        HIPCHK(hipEventElapsedTime(&c->ktimes[3], c->ev_lp, c->ev[6]));
    }
    for (int i = 0; i < last_ev; i++) {
        HIPCHK(hipEventElapsedTime(&c->timings[i], c->ev[i], c->ev[i + 1]));
        if (c->timings[i] < 0) c->timings[i] = 0;              // stages that ran side by side on the fork stream
    }
    if (last_ev == 7) {                           // batch path: the signature side and the lines by their own start events (forked calls)
        HIPCHK(hipEventElapsedTime(&c->timings[3], c->ev_s0, c->ev[4]));
        HIPCHK(hipEventElapsedTime(&c->timings[4], c->ev_l0, c->ev[5]));
    }
    HIPCHK(hipEventElapsedTime(&c->timings[7], c->ev[0], c->ev[last_ev]));
    return 0;
}

// Enqueue a whole batch verification (nothing is waited for); the verdict lands in the context's pinned host words.
static int verify_enqueue(mi355_bls_ctx* c, const uint8_t* d_sets, const uint8_t* h_sets, size_t n, const uint8_t rnd[32], int serial, hipStream_t st) {
    if (!c || !rnd) return MI355_BLS_ERR_ARG;
    if ((!d_sets && !h_sets) || n == 0) return MI355_BLS_ERR_ARG;
    if (c->pending) {
        g_err = "a batch submitted on this context has not been waited for";
        return MI355_BLS_ERR_ARG;
    }
    uint32_t B = (uint32_t)(n < c->num_threads ? n : c->num_threads);
    int rc = run_shard(c, d_sets, h_sets, n, B, 0, serial ? 1 : B, 0, n, serial, rnd, st);
    if (rc) return rc;
    k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, 2, c->d_gt, c->d_flags + 1, 144, 0);
    HIPCHK(hipEventRecord(c->ev[8], st));
    HIPCHK(hipMemcpyAsync(c->h_flags, c->d_flags, 8, hipMemcpyDeviceToHost, st));
    c->pending = true;
    c->pending_stream = st;
    return 0;
}
static int verify_wait(mi355_bls_ctx* c) {
    if (!c || !c->pending) return MI355_BLS_ERR_ARG;
    c->pending = false;
    HIPCHK(hipSetDevice(c->device));
    {
        hipStream_t ps = c->pending_stream;
        c->pending_stream = nullptr;              // the host may destroy its stream after this call: never keep the handle
        HIPCHK(hipStreamSynchronize(ps));
    }
    c->have_gt = true;
    float fin = 0;
    int rc = collect_timings(c, 7);
    if (rc) return rc;
    HIPCHK(hipEventElapsedTime(&fin, c->ev[7], c->ev[8]));
    c->timings[6] += fin;
    c->timings[7] += fin;
    return (c->h_flags[0] == 0 && c->h_flags[1] == 1) ? 1 : 0;
}
static int verify_common(mi355_bls_ctx* c, const uint8_t* d_sets, const uint8_t* h_sets, size_t n, const uint8_t rnd[32], int serial, hipStream_t st) {
    if (!c || !rnd) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;                      // bls_batch_verifier.nim:137-139, :312-314
    int rc = verify_enqueue(c, d_sets, h_sets, n, rnd, serial, st);
    if (rc) return rc;
    return verify_wait(c);
}

// where a batch submitted with `after` starts: behind that batch's hashing and public-key multiplications (ev[3]).
// MI355_BLS_CHAIN_EV = hm | clear | pk | sig | lines | lp moves the point (experiments: tools/abn.sh).
static hipEvent_t chain_event(mi355_bls_ctx* a) {
    static const int which = [] {
        const char* e = getenv("MI355_BLS_CHAIN_EV");
        if (!e) return 2;
        const char* names[] = {"hm", "clear", "pk", "sig", "lines", "lp"};
        for (int i = 0; i < 6; i++)
            if (!strcmp(e, names[i])) return i;
        return 2;
    }();
    switch (which) {
        case 0: return a->ev_hm;
        case 1: return a->ev[2];
        case 3: return a->ev[4];
        case 4: return a->ev[5];
        case 5: return a->ev_lp;
        default: return a->ev[3];
    }
}
extern "C" int mi355_bls_batch_submit_device(mi355_bls_ctx* c, const void* d_sets, size_t n, const uint8_t rnd[32], void* stream, mi355_bls_ctx* after) {
    if (after && after != c && after->wide_recorded) {
        // software pipelining: this batch starts when `after`'s batch has finished hashing and its public-key multiplications (the best of the
        // stage boundaries tried: 13.3 ms per batch against 13.7 one stage earlier and 17 one later), so the batches in flight sit
        // at different stages and the serial tail of one always runs beside whole-chip kernels of another (batches that
        // start together stay in phase: their tails coincide and leave the chip idle)
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, chain_event(after), 0));
    }
    return verify_enqueue(c, (const uint8_t*)d_sets, nullptr, n, rnd, 0, (hipStream_t)stream);
}
extern "C" int mi355_bls_batch_wait(mi355_bls_ctx* c) { return verify_wait(c); }

extern "C" int mi355_bls_batch_verify_device(mi355_bls_ctx* c, const void* d_sets, size_t n, const uint8_t rnd[32], void* stream) {
    return verify_common(c, (const uint8_t*)d_sets, nullptr, n, rnd, 0, (hipStream_t)stream);
}

static int verify_host(mi355_bls_ctx* c, const void* sets, size_t n, const uint8_t rnd[32], int serial) {
    if (!c || !rnd) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!sets) return MI355_BLS_ERR_ARG;
    return verify_common(c, nullptr, (const uint8_t*)sets, n, rnd, serial, nullptr);      // staged through d_sets, slice by slice when n exceeds the capacity
}

extern "C" int mi355_bls_batch_verify(mi355_bls_ctx* c, const void* sets, size_t n, const uint8_t rnd[32]) { return verify_host(c, sets, n, rnd, 0); }
extern "C" int mi355_bls_batch_verify_serial(mi355_bls_ctx* c, const void* sets, size_t n, const uint8_t rnd[32]) { return verify_host(c, sets, n, rnd, 1); }

// ------------------------------------------------------------------------------------------
// Many independent batches in ONE device pass.  A host that verifies many SMALL batches (a few thousand sets each: one per block or
// gossip aggregate) cannot fill the chip with one of them, and the number of HIP hardware queues caps how many calls run side by
// side ("Other rows").  Here the k batches are verified as the union of their tuples with every tuple keeping
// the blinding scalar it has in its OWN batch (its own secureRandomBytes, its own chain partition, batchVerify's dispatch rule per
// batch): the merged product is the product of the k batch products, so it is one iff every batch verifies - up to the 2^-64 of
// the random linear combination, which is the reference's own soundness bound for ONE batch.  If the merged check passes, every
// verdict is true (the common case, at whole-chip throughput); if it fails, the batches are verified one by one to find the
// culprits - the optimistic scheme clients already wrap around batchVerify.  Verdicts are exactly those of k separate calls.
// ------------------------------------------------------------------------------------------
static int verify_many(mi355_bls_ctx* c, const uint8_t* d_src, const uint8_t* h_src, const size_t counts[], const uint8_t* rnds, size_t k, uint8_t verdicts[],
                       hipStream_t st) {
    if (!c || !counts || !rnds || !verdicts || (!d_src && !h_src)) return MI355_BLS_ERR_ARG;
    if (c->pending) {
        g_err = "a batch submitted on this context has not been waited for";
        return MI355_BLS_ERR_ARG;
    }
    size_t total = 0;
    for (size_t b = 0; b < k; b++) {
        verdicts[b] = 0;
        total += counts[b];
    }
    if (total == 0) return 0;                                     // every batch empty: every verdict false (bls_batch_verifier.nim:137-139)
    HIPCHK(hipSetDevice(c->device));
    bool merged_ok = false;
    // The merged check is sound only if the batches' blinding scalars are independent.  They are a deterministic SHA-256 chain of
    // (rnd_b, chain id): two batches with the SAME secureRandomBytes and the same count get identical r_i at identical indices, and a
    // forger who knows that can make errors cancel across them (sig + D at index i of one, sig' - D at index i of the other: the
    // merged product is 1, both verdicts would be true, while k separate calls reject both).  A host that reuses one rnd for all its
    // batches is a plausible mistake and harmless with separate calls, so it must be harmless here: any two equal rnds among the
    // non-empty batches -> no merged pass, the batches are verified one by one.
    bool rnds_distinct = true;
    {
        std::vector<const uint8_t*> rs;
        for (size_t b = 0; b < k; b++)
            if (counts[b]) rs.push_back(rnds + 32 * b);
        std::sort(rs.begin(), rs.end(), [](const uint8_t* x, const uint8_t* y) { return memcmp(x, y, 32) < 0; });
        for (size_t i = 1; i < rs.size(); i++)
            if (memcmp(rs[i - 1], rs[i], 32) == 0) rnds_distinct = false;
    }
    if (rnds_distinct && total <= c->cap && k <= 65536) {
        // ---- merged pass
        std::vector<many_meta> meta;
        std::vector<uint8_t> rr;
        uint32_t lanes = 0;
        size_t first = 0;
        c->h_r.assign(total, 0);
        bool any_serial = false;
        for (size_t b = 0; b < k; b++) {
            size_t nb_ = counts[b];
            if (nb_ == 0) continue;                               // an empty batch is false by itself and takes no part
            const bool parallel = c->num_threads > 1 && nb_ >= 3; // batchVerify's dispatch (bls_batch_verifier.nim:440)
            if (parallel) {
                uint32_t B = (uint32_t)(nb_ < c->num_threads ? nb_ : c->num_threads);
                meta.push_back(many_meta{first, nb_, B, lanes});
                rr.insert(rr.end(), rnds + 32 * b, rnds + 32 * b + 32);
                lanes += B;
            } else {
                host_serial_chain(rnds + 32 * b, nb_, c->h_r.data() + first);
                any_serial = true;
            }
            first += nb_;
        }
        {   // meta and the per-batch random bytes ride in the wire-format staging buffer (unused on this path)
            int rcr = io_reserve(c, (meta.size() * (sizeof(many_meta) + 32) + 64 + 319) / 320 + 1);
            if (rcr) return rcr;
        }
        const uint8_t* d_sets = d_src;
        if (!d_sets) {
            HIPCHK(hipMemcpyAsync(c->d_sets, h_src, total * 320, hipMemcpyHostToDevice, st));
            d_sets = c->d_sets;
        }
        HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
        HIPCHK(hipEventRecord(c->ev[0], st));
        if (any_serial) HIPCHK(hipMemcpyAsync(c->d_r, c->h_r.data(), total * 8, hipMemcpyHostToDevice, st));      // serial batches' scalars (zeros elsewhere, overwritten below)
        if (lanes) {
            many_meta* d_meta = reinterpret_cast<many_meta*>(c->d_comp);
            uint8_t* d_rr = c->d_comp + ((meta.size() * sizeof(many_meta) + 63) / 64) * 64;
            HIPCHK(hipMemcpyAsync(d_meta, meta.data(), meta.size() * sizeof(many_meta), hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(d_rr, rr.data(), rr.size(), hipMemcpyHostToDevice, st));
            k_blind_many<<<(lanes + WAVE - 1) / WAVE, WAVE, 0, st>>>(d_rr, d_meta, (uint32_t)meta.size(), lanes, c->d_r);
        }
        int rc = run_pairs(c, d_sets, total, st);
        if (rc) return rc;
        k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, 2, c->d_gt, c->d_flags + 1, 144, 0);
        HIPCHK(hipEventRecord(c->ev[8], st));
        uint32_t fl[2];
        HIPCHK(hipMemcpyAsync(fl, c->d_flags, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));                         // also: meta / rr / h_r (host vectors) have been consumed
        c->have_gt = true;
        c->gt_is_fv = false;
        (void)collect_timings(c, 7);
        merged_ok = fl[0] == 0 && fl[1] == 1;
    }
    if (merged_ok) {
        int all = 1;
        for (size_t b = 0; b < k; b++) {
            verdicts[b] = counts[b] ? 1 : 0;
            all &= verdicts[b];
        }
        return all;
    }
    // ---- some batch fails (or the union exceeds the workspace): one by one, exactly as k separate batchVerify calls
    int all = 1;
    size_t first = 0;
    for (size_t b = 0; b < k; b++) {
        size_t nb_ = counts[b];
        int v = 0;
        if (nb_) {
            const int serial = (c->num_threads > 1 && nb_ >= 3) ? 0 : 1;
            v = verify_common(c, d_src ? d_src + 320 * first : nullptr, d_src ? nullptr : h_src + 320 * first, nb_, rnds + 32 * b, serial, st);
            if (v < 0) return v;
        }
        verdicts[b] = (uint8_t)v;
        all &= v;
        first += nb_;
    }
    return all;
}
extern "C" int mi355_bls_batch_verify_many(mi355_bls_ctx* c, const void* sets, const size_t counts[], const uint8_t* rnds, size_t k, uint8_t verdicts[]) {
    return verify_many(c, nullptr, (const uint8_t*)sets, counts, rnds, k, verdicts, nullptr);
}
extern "C" int mi355_bls_batch_verify_many_device(mi355_bls_ctx* c, const void* d_sets, const size_t counts[], const uint8_t* rnds, size_t k, uint8_t verdicts[],
                                                  void* stream) {
    return verify_many(c, (const uint8_t*)d_sets, nullptr, counts, rnds, k, verdicts, (hipStream_t)stream);
}

static int shard_enqueue(mi355_bls_ctx* c, const void* d_sets, const uint8_t* h_sets, size_t n_total, uint32_t chunk_lo, uint32_t chunk_hi, const uint8_t rnd[32],
                         hipStream_t st, mi355_bls_ctx* after) {
    if (!c || !rnd || n_total == 0 || (!d_sets && !h_sets)) return MI355_BLS_ERR_ARG;
    if (c->pending) {
        g_err = "a batch submitted on this context has not been waited for";
        return MI355_BLS_ERR_ARG;
    }
    uint32_t B = (uint32_t)(n_total < c->num_threads ? n_total : c->num_threads);
    if (chunk_hi > B) chunk_hi = B;
    if (chunk_lo >= chunk_hi) return MI355_BLS_ERR_ARG;
    size_t first, count;
    mi355_bls_chunk_range(n_total, c->num_threads, chunk_lo, chunk_hi, &first, &count);
    if (after && after != c && after->wide_recorded) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipStreamWaitEvent(st, after->ev[3], 0));          // see mi355_bls_batch_submit_device
    }
    int rc = run_shard(c, (const uint8_t*)d_sets, h_sets, n_total, B, chunk_lo, chunk_hi - chunk_lo, first, count, 0, rnd, st);
    if (rc) return rc;
    k_pack_blob<<<1, WAVE, 0, st>>>(c->d_states, c->d_flags, c->d_blob_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_flags + 16, c->d_states, 576, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_flags, c->d_flags, 4, hipMemcpyDeviceToHost, st));
    c->pending = true;
    c->pending_stream = st;
    return 0;
}
static int shard_wait(mi355_bls_ctx* c, uint8_t out_fp12[576], int* out_ok) {
    if (!c || !out_fp12 || !out_ok || !c->pending) return MI355_BLS_ERR_ARG;
    c->pending = false;
    HIPCHK(hipSetDevice(c->device));
    {
        hipStream_t ps = c->pending_stream;
        c->pending_stream = nullptr;
        HIPCHK(hipStreamSynchronize(ps));
    }
    std::memcpy(out_fp12, c->h_flags + 16, 576);
    *out_ok = c->h_flags[0] == 0 ? 1 : 0;
    return collect_timings(c, 7);
}
extern "C" int mi355_bls_batch_shard_device(mi355_bls_ctx* c, const void* d_sets, size_t n_total, uint32_t chunk_lo, uint32_t chunk_hi,
                                            const uint8_t rnd[32], void* stream, uint8_t out_fp12[576], int* out_ok) {
    if (!out_fp12 || !out_ok) return MI355_BLS_ERR_ARG;
    int rc = shard_enqueue(c, d_sets, nullptr, n_total, chunk_lo, chunk_hi, rnd, (hipStream_t)stream, nullptr);
    if (rc) return rc;
    return shard_wait(c, out_fp12, out_ok);
}
extern "C" int mi355_bls_batch_shard_submit_device(mi355_bls_ctx* c, const void* d_sets, size_t n_total, uint32_t chunk_lo, uint32_t chunk_hi,
                                                   const uint8_t rnd[32], void* stream, mi355_bls_ctx* after) {
    return shard_enqueue(c, d_sets, nullptr, n_total, chunk_lo, chunk_hi, rnd, (hipStream_t)stream, after);
}
extern "C" int mi355_bls_batch_shard_wait(mi355_bls_ctx* c, uint8_t out_fp12[576], int* out_ok) { return shard_wait(c, out_fp12, out_ok); }

extern "C" int mi355_bls_finalverify_shards(mi355_bls_ctx* c, const uint8_t* fp12s, size_t kk) {
    if (!c || !fp12s || kk == 0 || kk > 64) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_states, fp12s, kk * 576, hipMemcpyHostToDevice, nullptr));
    k_tail<<<1, tail_threads(c), 0, nullptr>>>(c->d_L, c->d_states, (uint32_t)kk, 2, c->d_gt, c->d_flags + 1, 144, 0);
    HIPCHK(hipGetLastError());
    uint32_t v = 0;
    HIPCHK(hipMemcpyAsync(&v, c->d_flags + 1, 4, hipMemcpyDeviceToHost, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    c->have_gt = true;
    c->gt_is_fv = false;
    return v == 1 ? 1 : 0;
}

extern "C" int mi355_bls_ctx_shard_blob_device(mi355_bls_ctx* c, void** d_blob) {
    if (!c || !d_blob) return MI355_BLS_ERR_ARG;
    *d_blob = c->d_blob_out;
    return 0;
}
extern "C" int mi355_bls_ctx_set_shard_blob_device(mi355_bls_ctx* c, void* d_blob) {
    if (!c || ((uintptr_t)d_blob & 15)) return MI355_BLS_ERR_ARG;
    c->d_blob_out = d_blob ? (uint32_t*)d_blob : c->d_blob;
    return 0;
}

// merge + finalVerify on k shard blobs resident in DEVICE memory (e.g. the output of an RCCL all_gather of every rank's
// mi355_bls_ctx_shard_blob_device buffer): nothing crosses PCIe but the verdict word.
extern "C" int mi355_bls_finalverify_blobs_submit_device(mi355_bls_ctx* c, const void* d_blobs, size_t kk, size_t stride_bytes, void* stream) {
    if (!c || !d_blobs || kk == 0 || kk > 1024 || stride_bytes < 580 || (stride_bytes & 3)) return MI355_BLS_ERR_ARG;
    if (c->fv_pending) {
        g_err = "a finalverify submitted on this context has not been waited for";
        return MI355_BLS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(d_blobs)), (uint32_t)kk, 2, c->d_gt_fv, c->d_flags + 3,
                               (uint32_t)(stride_bytes / 4), 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_flags + 3, c->d_flags + 3, 4, hipMemcpyDeviceToHost, st));
    c->fv_pending = true;
    c->fv_stream = st;
    return 0;
}
extern "C" int mi355_bls_finalverify_wait(mi355_bls_ctx* c) {
    if (!c || !c->fv_pending) return MI355_BLS_ERR_ARG;
    c->fv_pending = false;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->fv_stream));
    c->have_gt = true;
    c->gt_is_fv = true;
    return c->h_flags[3] == 1 ? 1 : 0;
}

// Contiguous, balanced blocks of chunks per device (the +-1 rule parallel_chunks uses for tuples, applied to chunks).
extern "C" int mi355_bls_shard_plan(size_t n_total, uint32_t num_threads, uint32_t world, uint32_t rank, uint32_t* chunk_lo, uint32_t* chunk_hi,
                                    size_t* first, size_t* count) {
    if (world == 0 || rank >= world || num_threads == 0 || !chunk_lo || !chunk_hi || !first || !count) return MI355_BLS_ERR_ARG;
    uint32_t B = (uint32_t)(n_total < num_threads ? n_total : num_threads);
    uint32_t base = B / world, rem = B % world;
    uint32_t lo = rank < rem ? (base + 1) * rank : base * rank + rem, hi = lo + base + (rank < rem ? 1 : 0);
    *chunk_lo = lo;
    *chunk_hi = hi;
    if (lo < hi) {
        mi355_bls_chunk_range(n_total, num_threads, lo, hi, first, count);
    } else {                                                  // more devices than chunks: an empty shard at the end of the batch
        size_t f0;
        mi355_bls_chunk_range(n_total, num_threads, 0, lo, &f0, first);
        *count = 0;
    }
    return 0;
}

// batchVerifyParallel over several GPUs from ONE host thread (bls_batch_verifier.nim:296-371 with devices in place of threads):
// device g takes a contiguous block of chunks (its processSingleChunk work, :326-357), all shards are enqueued asynchronously,
// the 576-byte committed states come back through pinned host memory, and device 0 merges them and runs the one final
// exponentiation (:360-371).  d_sets[g] != nullptr: shard g's records are already resident on device g.
// host time (us after the call began) at which each device's shard was handed to its stream in the last multi-device call of this thread
static thread_local float g_multi_enq_us[64];
static thread_local size_t g_multi_enq_n = 0;
extern "C" size_t mi355_bls_debug_multi_enqueue_us(float* out, size_t cap) {
    size_t k = g_multi_enq_n < cap ? g_multi_enq_n : cap;
    for (size_t i = 0; i < k; i++) out[i] = g_multi_enq_us[i];
    return g_multi_enq_n;
}
extern "C" int mi355_bls_debug_fail_next_enqueue(mi355_bls_ctx* c) {
    if (!c) return MI355_BLS_ERR_ARG;
    c->fail_next_enqueue = true;
    return 0;
}
static int verify_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, const uint8_t* sets, const void* const d_sets[], size_t n, const uint8_t rnd[32]) {
    if (!ctxs || ngpu == 0 || ngpu > 64 || !rnd || (!sets && !d_sets)) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    // first pass: the whole plan is validated before anything is enqueued, so that no argument error can strand a live shard
    struct plan_t { uint32_t lo, hi; size_t first, count; } plan[64];
    for (size_t g = 0; g < ngpu; g++) {
        if (!ctxs[g] || ctxs[g]->num_threads != ctxs[0]->num_threads) return MI355_BLS_ERR_ARG;
        if (ctxs[g]->pending) {
            g_err = "a batch submitted on this context has not been waited for";
            return MI355_BLS_ERR_ARG;
        }
        mi355_bls_shard_plan(n, ctxs[0]->num_threads, (uint32_t)ngpu, (uint32_t)g, &plan[g].lo, &plan[g].hi, &plan[g].first, &plan[g].count);
        if (plan[g].count && !(d_sets && d_sets[g]) && !sets) return MI355_BLS_ERR_ARG;
    }
    // Host records: the caller's range is page-locked for the duration of the call, so that every device's copy is a real
    // asynchronous DMA and device g does not wait for device g - 1's staging (from pageable memory hipMemcpyAsync blocks the host
    // thread until the copy has been staged: 42 MB per 131 072-tuple shard).  If the range cannot be registered the copies are
    // simply synchronous.
    bool registered = false;
    if (sets) {
        bool any_host = false;
        for (size_t g = 0; g < ngpu; g++) any_host = any_host || (plan[g].count && !(d_sets && d_sets[g]));
        if (any_host) {
            registered = hipHostRegister(const_cast<uint8_t*>(sets), n * 320, hipHostRegisterPortable) == hipSuccess;
            if (!registered) (void)hipGetLastError();
        }
    }
    bool live[64] = {};
    int rc_keep = 0;
    g_multi_enq_n = 0;
    const auto t_start = std::chrono::steady_clock::now();
    for (size_t g = 0; g < ngpu && !rc_keep; g++) {
        g_multi_enq_us[g_multi_enq_n++] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t_start).count();
        if (plan[g].count == 0) continue;                          // more devices than chunks
        mi355_bls_ctx* c = ctxs[g];
        const void* src = d_sets ? d_sets[g] : nullptr;
        // shard_enqueue stages host records itself (slice by slice when the shard exceeds the context's capacity)
        int rc = shard_enqueue(c, src, src ? nullptr : sets + 320 * plan[g].first, n, plan[g].lo, plan[g].hi, rnd, nullptr, nullptr);
        if (rc) rc_keep = rc;                                      // the shards already enqueued are waited for below
        else live[g] = true;
    }
    std::string err_keep = g_err;
    std::vector<uint8_t> states;
    bool all_ok = true;
    for (size_t g = 0; g < ngpu; g++) {
        if (!live[g]) continue;
        uint8_t st[576];
        int ok = 0;
        int rc = shard_wait(ctxs[g], st, &ok);                    // every enqueued shard is waited for, also after a failure
        if (rc && !rc_keep) {
            rc_keep = rc;
            err_keep = g_err;
        }
        all_ok = all_ok && ok;
        states.insert(states.end(), st, st + 576);
    }
    if (registered) (void)hipHostUnregister(const_cast<uint8_t*>(sets));
    if (rc_keep) {
        g_err = err_keep;
        return rc_keep;
    }
    if (!all_ok) return 0;                                        // some update() failed (infinity public key)
    return mi355_bls_finalverify_shards(ctxs[0], states.data(), states.size() / 576);
}
extern "C" int mi355_bls_batch_verify_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* sets, size_t n, const uint8_t rnd[32]) {
    if (n && !sets) return MI355_BLS_ERR_ARG;
    return verify_multi(ctxs, ngpu, (const uint8_t*)sets, nullptr, n, rnd);
}
extern "C" int mi355_bls_batch_verify_multi_device(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* const d_sets[], size_t n, const uint8_t rnd[32]) {
    if (n && !d_sets) return MI355_BLS_ERR_ARG;
    return verify_multi(ctxs, ngpu, nullptr, d_sets, n, rnd);
}

// Process-wide default context for the entry points that take no context (the reference's cache-less overloads allocate a
// cache per call, bls_batch_verifier.nim:399-416, :475-495; blst_p1s_mult_pippenger takes only a scratch pointer): created on
// first use on HIP device $MI355_BLS_DEVICE (default 0), regrown when a call needs more capacity; calls are serialised.
static std::mutex g_default_mu;
static mi355_bls_ctx* g_default_ctx = nullptr;
static int default_ctx_locked(size_t need_sets, mi355_bls_ctx** out) {
    if (need_sets < 1024) need_sets = 1024;
    // every entry point is capacity-free (a larger batch runs in slices, run_shard), so the default context never grows beyond two
    // whole-chip batches: mi355_bls_batch_verify_once on 2^20 sets allocates ~4 GB of workspace, not ~30
    if (need_sets > 131072) need_sets = 131072;
    if (g_default_ctx && g_default_ctx->cap >= need_sets) {
        *out = g_default_ctx;
        return 0;
    }
    if (g_default_ctx) {
        mi355_bls_ctx_destroy(g_default_ctx);
        g_default_ctx = nullptr;
    }
    const char* e = getenv("MI355_BLS_DEVICE");
    int rc = mi355_bls_ctx_create(&g_default_ctx, e ? atoi(e) : 0, need_sets);
    if (rc) return rc;
    *out = g_default_ctx;
    return 0;
}
extern "C" int mi355_bls_batch_verify_once(const void* sets, size_t n, const uint8_t rnd[32], uint32_t num_threads) {
    if (!rnd || num_threads == 0) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!sets) return MI355_BLS_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_default_mu);
    mi355_bls_ctx* c;
    int rc = default_ctx_locked(n, &c);
    if (rc) return rc;
    c->num_threads = num_threads;
    // batchVerify's dispatch (bls_batch_verifier.nim:475-495): parallel iff numThreads > 1 and n >= 3
    return verify_host(c, sets, n, rnd, (num_threads > 1 && n >= 3) ? 0 : 1);
}
extern "C" void mi355_bls_default_ctx_release(void) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default_ctx) mi355_bls_ctx_destroy(g_default_ctx);
    g_default_ctx = nullptr;
}

extern "C" int mi355_bls_fetch_stage(mi355_bls_ctx* c, int what, void* out, size_t out_bytes) {
    if (!c || !out) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    size_t n = c->last_n;
    uint32_t nb = (uint32_t)((n + 63) / 64);
    switch (what) {
        case 0:
            if (out_bytes < n * 8) return MI355_BLS_ERR_ARG;
            HIPCHK(hipMemcpy(out, c->d_r, n * 8, hipMemcpyDeviceToHost));
            return 0;
        case 1:
            if (out_bytes < n * 288 || n == 0) return MI355_BLS_ERR_ARG;
            k_export_g2<<<nb, 64>>>(c->d_H, c->stride, (uint32_t)n, c->d_export);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy(out, c->d_export, n * 288, hipMemcpyDeviceToHost));
            return 0;
        case 2:
            if (out_bytes < n * 144 || n == 0) return MI355_BLS_ERR_ARG;
            k_export_g1<<<nb, 64>>>(c->d_P, c->stride, (uint32_t)n, c->d_export);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpy(out, c->d_export, n * 144, hipMemcpyDeviceToHost));
            return 0;
        case 3:
            if (out_bytes < 288) return MI355_BLS_ERR_ARG;
            if (!c->agg_valid && c->sig_slots) {                     // bucket path: fold the bucket sums now
                k_sig_fold<<<1, WAVE>>>(c->d_H, c->stride, (uint32_t)n, 64 / c->sig_c, c->sig_c, c->d_agg);
                HIPCHK(hipGetLastError());
                c->agg_valid = true;
            }
            HIPCHK(hipMemcpy(out, c->d_agg, 288, hipMemcpyDeviceToHost));
            return 0;
        case 4:
            if (out_bytes < 576 || !c->have_gt) return MI355_BLS_ERR_ARG;
            HIPCHK(hipMemcpy(out, c->gt_is_fv ? c->d_gt_fv : c->d_gt, 576, hipMemcpyDeviceToHost));
            return 0;
        case 5:
            if (out_bytes < 576) return MI355_BLS_ERR_ARG;
            HIPCHK(hipMemcpy(out, c->d_states, 576, hipMemcpyDeviceToHost));
            return 0;
    }
    return MI355_BLS_ERR_ARG;
}

extern "C" int mi355_bls_last_kernel_timings(mi355_bls_ctx* c, float out[4]) {
    if (!c || !out) return MI355_BLS_ERR_ARG;
    for (int i = 0; i < 4; i++) out[i] = c->ktimes[i];
    return 0;
}

extern "C" int mi355_bls_last_timings(mi355_bls_ctx* c, float out[8]) {
    if (!c || !out) return MI355_BLS_ERR_ARG;
    for (int i = 0; i < 8; i++) out[i] = c->timings[i];
    return 0;
}

// ------------------------------------------------------------------------------------------
// aggregateAll / fastAggregateVerify
// ------------------------------------------------------------------------------------------
// aggregateAll on signatures (genAggregatorProcedures(AggregateSignature, Signature, p2), blst_min_pubkey_sig_core.nim:179-195,211):
// the same two-level sum over blst_p2_affine inputs (192 B), blst_p2 image out (288 B)
__global__ void __launch_bounds__(WAVE) k_g2_sum(const uint8_t* __restrict__ pts, uint32_t n, uint32_t m, uint32_t* __restrict__ part) {
    uint32_t lane0 = blockIdx.x * WAVE + threadIdx.x, strideL = gridDim.x * WAVE;
    g2_jac acc = jac_inf<fp2>();
#pragma clang loop unroll(disable)
    for (uint32_t j = 0; j < m; j++) {
        uint32_t i = lane0 + j * strideL;
        if (i < n) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(pts + (size_t)i * 192);
            acc = jac_add(acc, jac_from_aff(ld_g2a_blst(w)));
        }
    }
#pragma clang loop unroll(disable)
    for (int d = 32; d >= 1; d >>= 1) {
        g2_jac o = shfl_down_struct(acc, d);
        acc = jac_add(acc, o);
    }
    if (threadIdx.x == 0) st_g2_int(part + (size_t)blockIdx.x * G2W, acc);
}
__global__ void __launch_bounds__(WAVE) k_g2_sum2(const uint32_t* __restrict__ part, uint32_t nparts, uint32_t* __restrict__ out) {
    g2_jac acc = jac_inf<fp2>();
#pragma clang loop unroll(disable)
    for (uint32_t j = threadIdx.x; j < nparts; j += WAVE) acc = jac_add(acc, ld_g2_int(part + (size_t)j * G2W));
#pragma clang loop unroll(disable)
    for (int d = 32; d >= 1; d >>= 1) {
        g2_jac o = shfl_down_struct(acc, d);
        acc = jac_add(acc, o);
    }
    if (threadIdx.x == 0) st_g2_blst(out, acc);        // blst_p2 image
}
// one blst_p2 (Jacobian, 288 B) -> blst_p2_affine (192 B; infinity = all zero): finish(AggregateSignature) converts like this
// before the pairing (blst_min_pubkey_sig_core.nim:357-360: blst_p2_to_affine)
__global__ void k_p2_to_affine(const uint32_t* __restrict__ p2, uint32_t* __restrict__ out_sig) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    g2_jac b = ld_g2_blst(p2);
    if (jac_is_inf(b)) {
        for (int i = 0; i < 48; i++) out_sig[i] = 0;
    } else {
        fp2 zi = fp2_inv(fp2_reduce(b.z)), zi2 = fp2_sqr(zi);
        fp2 x = fp2_mul(b.x, zi2), y = fp2_mul(b.y, fp2_mul(zi2, zi));
        st_fp_blst(out_sig, x.c0); st_fp_blst(out_sig + 12, x.c1); st_fp_blst(out_sig + 24, y.c0); st_fp_blst(out_sig + 36, y.c1);
    }
}
static int g1_sum_enqueue(mi355_bls_ctx* c, const uint8_t* d_pts, size_t n, hipStream_t st) {
    // result (blst_p1 image, 144 B) lands in d_agg1
    uint32_t n32 = (uint32_t)n;
    uint32_t nblk = (n32 + WAVE * 8 - 1) / (WAVE * 8);          // ~8 points per lane
    if (nblk > c->slots * 2) nblk = c->slots * 2;
    if (nblk < 1) nblk = 1;
    uint32_t m = (n32 + nblk * WAVE - 1) / (nblk * WAVE);
    k_g1_sum<<<nblk, WAVE, 0, st>>>(d_pts, n32, m, c->d_export);
    k_g1_sum2<<<1, WAVE, 0, st>>>(c->d_export, nblk, c->d_agg1);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi355_bls_g1_aggregate_device(mi355_bls_ctx* c, const void* d_pks, size_t n, void* stream, uint8_t out_p1[144]) {
    if (!c || !d_pks || !out_p1 || n == 0 || n > (1u << 30)) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev[0], st));
    int rc = g1_sum_enqueue(c, (const uint8_t*)d_pks, n, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipMemcpyAsync(out_p1, c->d_agg1, 144, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    HIPCHK(hipEventElapsedTime(&c->timings[0], c->ev[0], c->ev[1]));
    c->timings[7] = c->timings[0];
    return 0;
}

extern "C" int mi355_bls_g2_aggregate_device(mi355_bls_ctx* c, const void* d_sigs, size_t n, void* stream, uint8_t out_p2[288]) {
    if (!c || !d_sigs || !out_p2 || n == 0 || n > (1u << 30)) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev[0], st));
    uint32_t n32 = (uint32_t)n;
    uint32_t nblk = (n32 + WAVE * 8 - 1) / (WAVE * 8);          // ~8 points per lane
    if (nblk > c->slots) nblk = c->slots;
    if (nblk > 2048) nblk = 2048;                               // d_export holds 2048 x 2 G1-sized partials beside its export area
    if (nblk < 1) nblk = 1;
    uint32_t m = (n32 + nblk * WAVE - 1) / (nblk * WAVE);
    k_g2_sum<<<nblk, WAVE, 0, st>>>((const uint8_t*)d_sigs, n32, m, c->d_export);
    k_g2_sum2<<<1, WAVE, 0, st>>>(c->d_export, nblk, c->d_agg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipMemcpyAsync(out_p2, c->d_agg, 288, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    HIPCHK(hipEventElapsedTime(&c->timings[0], c->ev[0], c->ev[1]));
    c->timings[7] = c->timings[0];
    return 0;
}
extern "C" int mi355_bls_g2_aggregate(mi355_bls_ctx* c, const void* sigs, size_t n, uint8_t out_p2[288]) {
    if (!c || !sigs || !out_p2 || n == 0) return MI355_BLS_ERR_ARG;
    {
        int rcr = io_reserve(c, (n * 192 + 319) / 320);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_sets, sigs, n * 192, hipMemcpyHostToDevice, nullptr));
    return mi355_bls_g2_aggregate_device(c, c->d_sets, n, nullptr, out_p2);
}
extern "C" int mi355_bls_g1_aggregate(mi355_bls_ctx* c, const void* pks, size_t n, uint8_t out_p1[144]) {
    if (!c || !pks || !out_p1 || n == 0) return MI355_BLS_ERR_ARG;
    {
        int rcr = io_reserve(c, (n * 96 + 319) / 320);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_sets, pks, n * 96, hipMemcpyHostToDevice, nullptr));
    return mi355_bls_g1_aggregate_device(c, c->d_sets, n, nullptr, out_p1);
}

// coreVerifyNoGroupCheck with the aggregate key (core :269-297).  d_pks != nullptr: the n keys are summed first (aggregateAll,
// beside the hash of the message in latency mode); d_pks == nullptr: the aggregate is already in d_agg1 (the multi-device form).
static int fav_run(mi355_bls_ctx* c, const void* d_pks, size_t n, const uint8_t* msg, size_t msg_len, const void* sig, hipStream_t st) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
    HIPCHK(hipMemcpyAsync(c->d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_msg + 4096, sig, 192, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(c->ev[0], st));
    // the key sum and the hash of the message are independent: side by side in latency mode
    hipStream_t sd = (d_pks && c->coop && c->side) ? c->side : st;
    if (sd != st) HIPCHK(hipStreamWaitEvent(sd, c->ev[0], 0));
    if (d_pks) {
        int rc = g1_sum_enqueue(c, (const uint8_t*)d_pks, n, sd);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(c->ev[1], sd));
    k_hash_one<<<1, WAVE, 0, st>>>(c->d_msg, (uint32_t)msg_len, c->dst, c->xmd, c->d_H, c->stride, 0);
    if (sd != st) HIPCHK(hipStreamWaitEvent(st, c->ev[1], 0));
    k_fav_setup<<<1, 1, 0, st>>>(c->d_agg1, reinterpret_cast<const uint32_t*>(c->d_msg + 4096), c->d_H, c->d_P, c->stride, c->d_flags);
    HIPCHK(hipEventRecord(c->ev[2], st));
    launch_lines(c, 2, 0, st);
    HIPCHK(hipEventRecord(c->ev[3], st));
    k_lineprod<<<dim3(N_LINES, 1), WAVE, 0, st>>>(c->d_lines, 2, c->stride, 1, c->d_lpart, 1, 0);
    k_lineprod2<<<N_LINES, WAVE, 0, st>>>(c->d_lpart, 1, c->d_L);
    HIPCHK(hipEventRecord(c->ev[4], st));
    k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, 3, c->d_gt, c->d_flags + 1, 144, 0);
    HIPCHK(hipEventRecord(c->ev[5], st));
    uint32_t fl[2];
    HIPCHK(hipMemcpyAsync(fl, c->d_flags, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    c->have_gt = true;
    c->gt_is_fv = false;
    c->last_n = 0;
    int rc = collect_timings(c, 5);      // [0] g1 sum, [1] hash+setup, [2] lines, [3] products, [4] tail
    if (rc) return rc;
    return (fl[0] == 0 && fl[1] == 1) ? 1 : 0;
}
extern "C" int mi355_bls_fast_aggregate_verify_device(mi355_bls_ctx* c, const void* d_pks, size_t n, const uint8_t* msg, size_t msg_len,
                                                      const void* sig, void* stream) {
    if (!c || !sig || (!msg && msg_len) || msg_len > 4096 || n > (1u << 30)) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;                                     // bls_sig_min_pubkey.nim:251-253
    if (!d_pks) return MI355_BLS_ERR_ARG;
    return fav_run(c, d_pks, n, msg, msg_len, sig, (hipStream_t)stream);
}

// coreVerifyNoGroupCheck on an aggregate the caller already holds (blst_min_pubkey_sig_core.nim:269-297 with an AggregatePublicKey:
// the `finish`-less form): agg_p1 = blst_p1 (Jacobian, 144 B), e.g. the sum of the per-rank partial key sums of a key-sharded
// fastAggregateVerify (mi355_bls_g1_aggregate_device per rank, mi355_bls_p1s_add on rank 0).  Aggregate at infinity -> 0.
extern "C" int mi355_bls_verify_aggregate(mi355_bls_ctx* c, const uint8_t agg_p1[144], const uint8_t* msg, size_t msg_len, const void* sig) {
    if (!c || !agg_p1 || !sig || (!msg && msg_len) || msg_len > 4096) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->d_agg1, agg_p1, 144, hipMemcpyHostToDevice));
    return fav_run(c, nullptr, 1, msg, msg_len, sig, nullptr);
}

// fastAggregateVerify with the keys sharded over several devices: device g sums keys [first_g, first_g +
// count_g) (mi355_bls_msm_shard_range), the 144-byte partial sums return through pinned host memory, ctxs[0] adds them and runs the
// one pairing check.  At 3 MB of keys one device is the sensible default; this is the same call for key sets that are not.
extern "C" int mi355_bls_fast_aggregate_verify_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, const void* pks, size_t n, const uint8_t* msg,
                                                     size_t msg_len, const void* sig) {
    if (!ctxs || ngpu == 0 || ngpu > 64 || !sig || (!msg && msg_len) || msg_len > 4096 || n > (1u << 30)) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!pks) return MI355_BLS_ERR_ARG;
    for (size_t g = 0; g < ngpu; g++)
        if (!ctxs[g]) return MI355_BLS_ERR_ARG;
    const uint8_t* p = (const uint8_t*)pks;
    bool live[64] = {};
    int rc = 0;
    for (size_t g = 0; g < ngpu && !rc; g++) {
        size_t first, count;
        mi355_bls_msm_shard_range(n, (uint32_t)ngpu, (uint32_t)g, &first, &count);
        if (count == 0) continue;
        mi355_bls_ctx* c = ctxs[g];
        rc = io_reserve(c, (count * 96 + 319) / 320);
        if (rc) break;
        if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(c->d_sets, p + 96 * first, count * 96, hipMemcpyHostToDevice, nullptr) != hipSuccess) {
            g_err = "staging of a key shard failed";
            rc = MI355_BLS_ERR_HIP;
            break;
        }
        rc = g1_sum_enqueue(c, c->d_sets, count, nullptr);
        if (rc) break;
        if (hipMemcpyAsync(c->h_flags + 160, c->d_agg1, 144, hipMemcpyDeviceToHost, nullptr) != hipSuccess) { g_err = "hipMemcpyAsync (key-sum partial)"; rc = MI355_BLS_ERR_HIP; break; }
        live[g] = true;
    }
    std::vector<uint8_t> parts;
    for (size_t g = 0; g < ngpu; g++) {                  // every device that was handed work is waited for, also after a failure
        if (!live[g]) continue;
        (void)hipSetDevice(ctxs[g]->device);
        if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) { g_err = "hipStreamSynchronize (key shard)"; rc = MI355_BLS_ERR_HIP; }
        const uint8_t* h = reinterpret_cast<const uint8_t*>(ctxs[g]->h_flags + 160);
        parts.insert(parts.end(), h, h + 144);
    }
    if (rc) return rc;
    mi355_bls_ctx* c0 = ctxs[0];
    HIPCHK(hipSetDevice(c0->device));
    HIPCHK(hipMemcpyAsync(c0->d_export, parts.data(), parts.size(), hipMemcpyHostToDevice, nullptr));
    k_jac_sum_blst<fp><<<1, WAVE, 0, nullptr>>>(c0->d_export, (uint32_t)(parts.size() / 144), 36, c0->d_agg1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));               // `parts` (pageable) has been consumed
    return fav_run(c0, nullptr, n, msg, msg_len, sig, nullptr);
}

extern "C" int mi355_bls_fast_aggregate_verify(mi355_bls_ctx* c, const void* pks, size_t n, const uint8_t* msg, size_t msg_len, const void* sig) {
    if (!c) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!pks) return MI355_BLS_ERR_ARG;
    {
        int rcr = io_reserve(c, (n * 96 + 319) / 320);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_sets, pks, n * 96, hipMemcpyHostToDevice, nullptr));
    return mi355_bls_fast_aggregate_verify_device(c, c->d_sets, n, msg, msg_len, sig, nullptr);
}

// ------------------------------------------------------------------------------------------
// blst_p1s_mult_pippenger / blst_p2s_mult_pippenger replacement (host side)
// ------------------------------------------------------------------------------------------
constexpr uint32_t MSM_SEG = 16;
// buckets per running-sum segment of k_pip_segred: shorter running sums once they still fill the chip.  MI355_BLS_MSM_SEG (4, 8, 16) overrides
// it for experiments (tools/msm_quick.sh).
static uint32_t msm_seg_len(uint32_t cbk) {
    static const uint32_t forced = [] {
        const char* e = getenv("MI355_BLS_MSM_SEG");
        uint32_t v = e ? (uint32_t)atoi(e) : 0;
        return (v == 4 || v == 8 || v == 16) ? v : 0u;
    }();
    if (forced && (1u << cbk) >= forced) return forced;
    return cbk >= 12 ? 8u : MSM_SEG;
}

// Window plan for npoints x nbits: about log2(n) - 3 bits per window (signed digits: 2^(c-1) buckets), widths balanced.
static pip_win pip_plan(size_t npoints, size_t nbits) {
    uint32_t lg = 0;
    while ((1ull << (lg + 1)) <= npoints) lg++;
    int c = (int)lg - 3;
    if (c < 5) c = 5;                                   // at least one 16-bucket segment per window
    if (c > 16) c = 16;                                 // at most 2^15 buckets per window: the counters of the LDS counting sort
    pip_win W{};
    W.nbits = (uint32_t)nbits;
    uint32_t ext = (uint32_t)nbits + 1;                  // one extra (zero) top bit: the top window absorbs the carry of the bias
    W.nwin = (ext + c - 1) / c;
    W.wbase = ext / W.nwin;
    W.wrem = ext % W.nwin;
    uint32_t widest = W.wbase + (W.wrem ? 1 : 0);
    W.cbk = widest - 1;
    if (W.cbk < 4) W.cbk = 4;
    for (int j = 0; j < 9; j++) W.H[j] = 0;
    for (uint32_t w = 0; w + 1 < W.nwin; w++) {
        uint32_t off = w < W.wrem ? w * (W.wbase + 1) : W.wrem * (W.wbase + 1) + (w - W.wrem) * W.wbase;
        uint32_t len = w < W.wrem ? W.wbase + 1 : W.wbase;
        uint32_t bit = off + len - 1;                    // + 2^(len - 1) at window w
        W.H[bit >> 5] |= 1u << (bit & 31);
    }
    return W;
}

extern "C" size_t mi355_bls_p1s_mult_pippenger_scratch_sizeof(size_t npoints) {
    (void)npoints;
    return 0;          // blst_p1s_mult_pippenger_scratch_sizeof (blst_abi.nim:336): the workspace lives on the device
}

// workspace for npoints points of `affb`-byte affine images (96: G1, 192: G2) under window plan W
static int msm_reserve(mi355_bls_ctx* c, msm_ws* m, size_t n, const pip_win& W, size_t affb) {
    (void)c;
    uint32_t total = W.nwin << W.cbk;
    size_t pts_bytes = n * affb;
    if (pts_bytes <= m->cap_n && total <= m->cap_total) return 0;
    size_t cb = pts_bytes > m->cap_n ? pts_bytes : m->cap_n;
    uint32_t ct = total > m->cap_total ? total : m->cap_total;
    msm_free(m);
#define MALLOC(p, bytes)                                                                   \
    do {                                                                                   \
        hipError_t e_ = hipMalloc((void**)&(p), (bytes));                                  \
        if (e_ != hipSuccess) {                                                            \
            g_err = std::string("hipMalloc " #p ": ") + hipGetErrorString(e_);             \
            msm_free(m);                                                                   \
            return MI355_BLS_ERR_HIP;                                                      \
        }                                                                                  \
    } while (0)
    size_t cn = cb / 96;                                 // point capacity counted in G1 points (a G2 point takes two)
    MALLOC(m->d_pts, cb);
    MALLOC(m->d_sc, cn * 32);
    MALLOC(m->pts_int, cn * 2 * FPW * 4);
    MALLOC(m->hist, (size_t)ct * 4);
    MALLOC(m->offs, (size_t)ct * 4);
    MALLOC(m->cursor, (size_t)ct * 4);
    MALLOC(m->order, (size_t)ct * 4);
    MALLOC(m->chist, 4 * 256 * 4);
    MALLOC(m->shist, (size_t)ct * PIP_SLICES * 4);      // per-slice counters of the LDS counting sort
    MALLOC(m->part, 64 * 16 * G2W * 4);                 // per window up to 16 partial sums
    MALLOC(m->sorted, (size_t)cn * 64 * 4);          // up to 52 + 1 windows (nbits 256 at 5-bit windows)
    MALLOC(m->buckets, (size_t)ct * 6 * 64);
    MALLOC(m->segout, (size_t)(ct / 4 + 64) * 6 * 64);
    MALLOC(m->winout, 64 * G2W * 4);
    MALLOC(m->out, 288);
#undef MALLOC
    HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_bucketed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    for (int g = 0; g < 4; g++) HIPCHK(hipEventCreateWithFlags(&m->ev_g[g], hipEventDisableTiming));
    HIPCHK(hipStreamCreateWithFlags(&m->gs3, hipStreamNonBlocking));
    m->cap_n = cb;
    m->cap_total = ct;
    return 0;
}

// Everything up to the result in m->out (blst_p1 / blst_p2 image, device memory) is ENQUEUED on `st` with workspace m; nothing is
// waited for.  timed: record the context's stage events.  allow_split: the window groups may use the context's side stream.
template <class F>
static int msm_enqueue(mi355_bls_ctx* c, msm_ws* m, const void* d_points, size_t npoints, const void* d_scalars, uint32_t sbytes, size_t nbits,
                       hipStream_t st, bool timed, bool allow_split) {
    constexpr size_t AFFB = sizeof(F) == sizeof(fp) ? 96 : 192;
    pip_win W = pip_plan(npoints, nbits);
    int rc = msm_reserve(c, m, npoints, W, AFFB);
    if (rc) return rc;
    uint32_t n = (uint32_t)npoints, nw = W.nwin, total = nw << W.cbk, seg = msm_seg_len(W.cbk), segs_per_win = (1u << W.cbk) / seg,      // shorter running sums once they still fill the chip
             nseg = nw * segs_per_win;
    const uint8_t* pts = (const uint8_t*)d_points;
    const uint8_t* sc = (const uint8_t*)d_scalars;
    uint32_t nbp = (n + WAVE - 1) / WAVE;
    uint32_t nsplit = segs_per_win >= 1024 ? 16 : (segs_per_win >= 128 ? 4 : 1);
    // One group of windows = the whole pipeline on a range of windows [w0, w1): sort -> buckets -> segment sums -> window sums.
    const bool lds_sort = W.cbk <= PIP_SORT_MAX_CBK && W.cbk >= 10 && n >= (1u << 15);        // counters of a window in LDS (large inputs)
    auto count_sort = [&](uint32_t w0, uint32_t w1, hipStream_t s) {
        uint32_t g0 = w0 << W.cbk, gc = (w1 - w0) << W.cbk;
        if (lds_sort) {
            uint32_t per = (n + PIP_SLICES - 1) / PIP_SLICES;
            k_pip_hist_lds<<<dim3(PIP_SLICES, w1 - w0), PIP_SORT_THREADS, 0, s>>>(sc, sbytes, n, W, w0, per, m->shist);
            k_pip_slice_scan<<<(gc + WAVE - 1) / WAVE, WAVE, 0, s>>>(m->shist, PIP_SLICES, W.cbk, g0, gc, m->hist);
            k_pip_scan_block<<<w1 - w0, PIP_SORT_THREADS, 0, s>>>(m->hist + g0, W.cbk, m->offs + g0);
            k_pip_scatter_lds<<<dim3(PIP_SLICES, w1 - w0), PIP_SORT_THREADS, 0, s>>>(sc, sbytes, n, W, w0, per, m->shist, m->offs, m->sorted);
        } else {
            k_pip_hist<<<dim3(nbp, w1 - w0), WAVE, 0, s>>>(sc, sbytes, n, W, w0, m->hist);
            k_msm_scan<<<w1 - w0, WAVE, 0, s>>>(m->hist + g0, W.cbk, m->offs + g0, m->cursor + g0);
            k_pip_scatter<<<dim3(nbp, w1 - w0), WAVE, 0, s>>>(sc, sbytes, n, W, w0, m->cursor, m->sorted);
        }
    };
    auto order_group = [&](uint32_t w0, uint32_t w1, uint32_t gi, hipStream_t s) {      // the group's buckets by load (indices relative to g0)
        uint32_t g0 = w0 << W.cbk, gc = (w1 - w0) << W.cbk, nbo = (gc + WAVE * MSM_ORD_PER - 1) / (WAVE * MSM_ORD_PER);
        uint32_t* chist = m->chist + 256 * gi;
        k_msm_order_hist<<<nbo, WAVE, 0, s>>>(m->hist + g0, gc, chist);
        k_msm_order_scan<<<1, 1, 0, s>>>(chist);
        k_msm_order_scatter<<<nbo, WAVE, 0, s>>>(m->hist + g0, gc, chist, m->order + g0);
    };
    auto bucket_group = [&](uint32_t w0, uint32_t w1, hipStream_t s) {
        uint32_t g0 = w0 << W.cbk, gc = (w1 - w0) << W.cbk;
        k_pip_bucket<F><<<(gc + WAVE - 1) / WAVE, WAVE, 0, s>>>(m->pts_int, m->sorted, m->offs, m->hist, m->order, n, W.cbk, total, g0, gc, m->buckets);
    };
    auto reduce_group = [&](uint32_t w0, uint32_t w1, hipStream_t s) {
        uint32_t t0 = w0 * segs_per_win, tc = (w1 - w0) * segs_per_win;
        // G1: 4 or 2 lanes per segment (lane teams) while the team waves stay well inside the chip's 1024 one-per-SIMD wave slots (<= 960 waves: a
        // kernel of exactly 1024 such waves finds a few SIMDs taken by the other group's reduction and runs a second round for the stragglers).
        // profiles/r04_ab/msm_team.txt: 2^14 points 2.70 -> 2.29 ms, 2^16 2.56 -> 2.28, 2^18 3.28 -> 3.10 (two lanes); 2^20 would need 1024 waves per
        // group and measured 5.2 - 5.4 ms against 5.1 - 5.2: one lane per segment there.  MI355_BLS_MSM_TEAM = 1 / 2 / 4 forces a size.
        static const int team_forced = getenv("MI355_BLS_MSM_TEAM") ? atoi(getenv("MI355_BLS_MSM_TEAM")) : 0;
        int team = sizeof(F) != sizeof(fp) ? 1 : (team_forced > 0 ? team_forced : ((size_t)tc * 4 <= 61440 ? 4 : ((size_t)tc * 2 <= 61440 ? 2 : 1)));
        if (team == 4) k_pip_segred_team<4><<<(tc * 4 + WAVE - 1) / WAVE, WAVE, 0, s>>>(m->buckets, total, W.cbk, seg, nseg, t0, tc, m->segout);
        else if (team == 2) k_pip_segred_team<2><<<(tc * 2 + WAVE - 1) / WAVE, WAVE, 0, s>>>(m->buckets, total, W.cbk, seg, nseg, t0, tc, m->segout);
        else k_pip_segred<F><<<(tc + WAVE - 1) / WAVE, WAVE, 0, s>>>(m->buckets, total, W.cbk, seg, nseg, t0, tc, m->segout);
        k_pip_winpart<F><<<dim3(w1 - w0, nsplit), WAVE, 0, s>>>(m->segout, nseg, segs_per_win, w0, m->part);
        k_pip_winsum<F><<<w1 - w0, WAVE, 0, s>>>(m->part, nsplit, W, w0, m->winout);
    };
    HIPCHK(hipMemsetAsync(m->hist, 0, (size_t)total * 4, st));
    HIPCHK(hipMemsetAsync(m->chist, 0, 4 * 256 * 4, st));
    if (timed) HIPCHK(hipEventRecord(c->ev[0], st));
    k_pip_convert<F><<<nbp, WAVE, 0, st>>>(pts, n, m->pts_int);
    // The counting sort covers all windows; then two groups of windows, each on its own stream: the HIGH windows first (their
    // results need the long doubling chains: up to nbits - c dependent doublings on one wave per window, ~1 ms of pure
    // latency), the LOW windows' bucket kernel behind the high one, so that the high group's serial tail runs beside the bucket
    // accumulation of the low group and only the short chains of the low windows are left at the end.  More groups lose more
    // in the bucket kernels' tails than they hide.  Large inputs only: a small MSM is latency-bound in every stage.
    const bool split = allow_split && c->side && nw >= 4 && (size_t)n * nw >= ((size_t)1 << 22);
    // Groups [cut[g + 1], cut[g]) from the high windows down: two halves.  MI355_BLS_MSM_CUTS="a" or "a,b" (window indices, descending)
    // moves the cut or makes three groups for experiments (tools/msm_cuts.sh).  Measured at 2^20 x 255 bits, 16 windows
    // (profiles/r04_ab/msm_cuts.txt): cuts 5 .. 10 are within the noise of 8; three groups (10,4 / 11,5 / 12,6 / 9,3), whose last
    // group's exposed reduction is shorter, are 1 - 3 % SLOWER alone and 10 % slower with two MSMs in flight - every extra group's
    // bucket kernel has its own tail and shares the chip with one more reduction.
    uint32_t cut[5] = {nw, 0, 0, 0, 0}, ngroups = 1;
    if (split) {
        static const char* e = getenv("MI355_BLS_MSM_CUTS");
        uint32_t a = nw / 2, b = 0;
        if (e) {
            a = (uint32_t)atoi(e);
            const char* q = strchr(e, ',');
            b = q ? (uint32_t)atoi(q + 1) : 0;
            if (a == 0 || a >= nw || b >= a) { a = nw / 2; b = 0; }
        }
        cut[1] = a;
        ngroups = 2;
        if (b) { cut[2] = b; ngroups = 3; }
    }
    // group g runs on its own stream, its bucket kernel behind the bucket kernel of group g - 1: the (latency-bound, few-wave)
    // reduction of a group is dispatched before the next group's bucket kernel and runs beside it.  (Both bucket kernels enqueued at
    // once, the second on a lowest-priority stream so that its waves would only fill the tail of the first - 26 % of a bucket
    // kernel's wave slots idle on average, profiles/r03_pmc_summary_msm.json - was measured 3 % SLOWER: the 512-register reduction
    // waves of the first group then wait for whole SIMDs that the second group's 256-register waves keep half full.)
    hipStream_t gs[3] = {st, c->side, m->gs3};
    count_sort(0, nw, st);
    for (uint32_t g = 0; g < ngroups; g++) order_group(cut[g + 1], cut[g], g, st);
    if (timed) HIPCHK(hipEventRecord(c->ev[1], st));
    if (ngroups > 2) {                                                   // the third stream starts behind everything enqueued so far
        HIPCHK(hipEventRecord(m->ev_join, st));
        HIPCHK(hipStreamWaitEvent(gs[2], m->ev_join, 0));
    }
    for (uint32_t g = 0; g < ngroups; g++) {
        if (g) HIPCHK(hipStreamWaitEvent(gs[g], m->ev_g[g - 1], 0));
        bucket_group(cut[g + 1], cut[g], gs[g]);
        HIPCHK(hipEventRecord(m->ev_g[g], gs[g]));
        if (g == 0 && timed) HIPCHK(hipEventRecord(c->ev[2], st));
        reduce_group(cut[g + 1], cut[g], gs[g]);
        if (g == 0 && timed) HIPCHK(hipEventRecord(c->ev[3], st));
    }
    for (uint32_t g = 1; g < ngroups; g++) {
        HIPCHK(hipEventRecord(m->ev_g[g], gs[g]));
        HIPCHK(hipStreamWaitEvent(st, m->ev_g[g], 0));
    }
    k_pip_final<F><<<1, WAVE, 0, st>>>(m->winout, nw, m->out);
    if (timed) HIPCHK(hipEventRecord(c->ev[4], st));
    HIPCHK(hipGetLastError());
    return 0;
}
// sum_i [k_i mod 2^nbits] P_i on the device.  sbytes: distance between scalars (32 for blst_scalar images; blst's own
// convention is (nbits + 7) / 8).  ret: blst_p1 (144 B) or blst_p2 (288 B), host memory.
template <class F>
static int msm_run(mi355_bls_ctx* c, uint8_t* ret, const void* d_points, size_t npoints, const void* d_scalars, uint32_t sbytes, size_t nbits, void* stream) {
    constexpr size_t JACB = (sizeof(F) == sizeof(fp) ? 96 : 192) / 2 * 3;
    if (!c || !ret || nbits == 0 || nbits > 256 || npoints > (1u << 28) || (size_t)sbytes * 8 < nbits || sbytes > 32) return MI355_BLS_ERR_ARG;
    if (npoints == 0) {
        memset(ret, 0, JACB);
        return 0;
    }
    if (!d_points || !d_scalars) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    int rc = msm_enqueue<F>(c, c->msm, d_points, npoints, d_scalars, sbytes, nbits, st, true, true);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ret, c->msm->out, JACB, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return collect_timings(c, 4);       // [0] sort, [1] bucket accumulation, [2] segment reduction, [3] window sums + doublings
}
extern "C" int mi355_bls_p1s_mult_pippenger_device(mi355_bls_ctx* c, uint8_t ret_p1[144], const void* d_points, size_t npoints, const void* d_scalars,
                                                   size_t nbits, void* stream) {
    return msm_run<fp>(c, ret_p1, d_points, npoints, d_scalars, 32, nbits, stream);
}
extern "C" int mi355_bls_p2s_mult_pippenger_device(mi355_bls_ctx* c, uint8_t ret_p2[288], const void* d_points, size_t npoints, const void* d_scalars,
                                                   size_t nbits, void* stream) {
    return msm_run<fp2>(c, ret_p2, d_points, npoints, d_scalars, 32, nbits, stream);
}

// host arrays (contiguous) -> staging -> msm_run
template <class F>
static int msm_host(mi355_bls_ctx* c, uint8_t* ret, const uint8_t* pts, size_t npoints, const uint8_t* scalars, uint32_t sbytes, size_t nbits) {
    constexpr size_t AFFB = sizeof(F) == sizeof(fp) ? 96 : 192;
    if (!c || nbits == 0 || nbits > 256 || npoints > (1u << 28)) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    int rc = msm_reserve(c, c->msm, npoints, pip_plan(npoints, nbits), AFFB);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->msm->d_pts, pts, npoints * AFFB, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(c->msm->d_sc, scalars, npoints * sbytes, hipMemcpyHostToDevice, nullptr));
    return msm_run<F>(c, ret, c->msm->d_pts, npoints, c->msm->d_sc, sbytes, nbits, nullptr);
}

// Same shape as blst_p1s_mult_pippenger incl. the NULL-terminated pointer-to-array convention
// (blst+nim.h:70-72; benchmarks/bls12381_msm_g1.nim:52-59), but with a context, an int result and 32-byte scalar images
// (blst_scalar arrays) whatever nbits is: points[0] / scalars[0] are contiguous arrays in HOST memory.
extern "C" int mi355_bls_p1s_mult_pippenger(mi355_bls_ctx* c, uint8_t ret_p1[144], const void* const points[], size_t npoints,
                                            const uint8_t* const scalars[], size_t nbits) {
    if (!c || !ret_p1) return MI355_BLS_ERR_ARG;
    if (npoints == 0) {
        memset(ret_p1, 0, 144);
        return 0;
    }
    if (!points || !points[0] || !scalars || !scalars[0] || nbits == 0 || nbits > 256) return MI355_BLS_ERR_ARG;
    return msm_host<fp>(c, ret_p1, (const uint8_t*)points[0], npoints, scalars[0], 32, nbits);
}
extern "C" int mi355_bls_p2s_mult_pippenger(mi355_bls_ctx* c, uint8_t ret_p2[288], const void* const points[], size_t npoints,
                                            const uint8_t* const scalars[], size_t nbits) {
    if (!c || !ret_p2) return MI355_BLS_ERR_ARG;
    if (npoints == 0) {
        memset(ret_p2, 0, 288);
        return 0;
    }
    if (!points || !points[0] || !scalars || !scalars[0] || nbits == 0 || nbits > 256) return MI355_BLS_ERR_ARG;
    return msm_host<fp2>(c, ret_p2, (const uint8_t*)points[0], npoints, scalars[0], 32, nbits);
}

// blst's list convention (blst_p1s_mult_pippenger and friends): list[0] points at element 0; for every following element the
// next list entry is used if it is non-NULL, otherwise the element follows the previous one in memory.  [ptr, NULL] is one
// contiguous array (what the reference passes, benchmarks/bls12381_msm_g1.nim:52-55, core :613-616); npoints pointers
// address every element individually.  Returns a contiguous view (gathered into tmp when needed).
static const uint8_t* gather_list(const void* const list[], size_t n, size_t elem, std::vector<uint8_t>& tmp) {
    const uint8_t* cur = (const uint8_t*)list[0];
    if (n <= 1 || list[1] == nullptr) return cur;
    tmp.resize(n * elem);
    std::memcpy(tmp.data(), cur, elem);
    size_t li = 1;
    for (size_t i = 1; i < n; i++) {
        if (list[li]) cur = (const uint8_t*)list[li++];
        else cur += elem;
        std::memcpy(tmp.data() + i * elem, cur, elem);
    }
    return tmp.data();
}
[[noreturn]] static void die_no_error_channel(const char* fn) {
    std::fprintf(stderr, "%s: %s (this entry point has blst's void signature, so a runtime failure cannot be returned; aborting rather than "
                 "handing back a wrong point)\n", fn, g_err.c_str());
    std::abort();
}

// EXACTLY blst_p1s_mult_pippenger / blst_p2s_mult_pippenger (blst+nim.h:70-72,90-92; blst_abi.nim:336-340,358-362): no context
// (the process-wide default one), void, scalars (nbits + 7) / 8 bytes apart, scratch ignored (the workspace lives on the device).
template <class F>
static void blst_shaped_pippenger(const char* fn, void* ret, const void* const points[], size_t npoints, const uint8_t* const scalars[], size_t nbits) {
    constexpr size_t AFFB = sizeof(F) == sizeof(fp) ? 96 : 192, JACB = AFFB / 2 * 3;
    if (!ret) return;
    if (npoints == 0) {
        std::memset(ret, 0, JACB);
        return;
    }
    if (!points || !points[0] || !scalars || !scalars[0] || nbits == 0 || nbits > 256) {
        g_err = "bad arguments";
        die_no_error_channel(fn);
    }
    std::lock_guard<std::mutex> lk(g_default_mu);
    mi355_bls_ctx* c;
    uint32_t sbytes = (uint32_t)((nbits + 7) / 8);
    std::vector<uint8_t> tp, ts;
    const uint8_t* P = gather_list(points, npoints, AFFB, tp);
    const uint8_t* S = gather_list(reinterpret_cast<const void* const*>(scalars), npoints, sbytes, ts);
    int rc = default_ctx_locked(1024, &c);
    if (!rc) rc = msm_host<F>(c, (uint8_t*)ret, P, npoints, S, sbytes, nbits);
    if (rc) die_no_error_channel(fn);
}
extern "C" size_t mi355_p1s_mult_pippenger_scratch_sizeof(size_t npoints) {
    (void)npoints;
    return 8;                       // never 0: callers malloc() it (benchmarks/bls12381_msm_g1.nim:50) and index scratch[0] (core :633)
}
extern "C" size_t mi355_p2s_mult_pippenger_scratch_sizeof(size_t npoints) {
    (void)npoints;
    return 8;
}
extern "C" void mi355_p1s_mult_pippenger(void* ret, const void* const points[], size_t npoints, const uint8_t* const scalars[], size_t nbits, void* scratch) {
    (void)scratch;
    blst_shaped_pippenger<fp>("mi355_p1s_mult_pippenger", ret, points, npoints, scalars, nbits);
}
extern "C" void mi355_p2s_mult_pippenger(void* ret, const void* const points[], size_t npoints, const uint8_t* const scalars[], size_t nbits, void* scratch) {
    (void)scratch;
    blst_shaped_pippenger<fp2>("mi355_p2s_mult_pippenger", ret, points, npoints, scalars, nbits);
}

// ------------------------------------------------------------------------------------------
// Wire-format entry points: batched fromBytes (+ batchVerify)
// ------------------------------------------------------------------------------------------
static int deser_enqueue(mi355_bls_ctx* c, const uint8_t* d_pks, const uint8_t* d_msgs, const uint8_t* d_sigs, size_t n, uint32_t dflags, hipStream_t st) {
    {
        int rcr = io_reserve(c, n);
        if (rcr) return rcr;
    }
    if (dflags > 7) return MI355_BLS_ERR_ARG;
    HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
    k_deser<<<((uint32_t)n + WAVE - 1) / WAVE, WAVE, 0, st>>>(d_pks, d_msgs, d_sigs, (uint32_t)n, dflags, c->d_sets, c->d_status, c->d_flags);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi355_bls_deserialize_sets_ex_device(mi355_bls_ctx* c, const void* d_pks48, const void* d_msgs32, const void* d_sigs96, size_t n, uint32_t dflags,
                                                    void* stream, void* out_sets, uint8_t* status) {
    if (!c) return MI355_BLS_ERR_ARG;
    if (n == 0) return 1;
    if (!d_pks48 || !d_msgs32 || !d_sigs96) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev[0], st));
    int rc = deser_enqueue(c, (const uint8_t*)d_pks48, (const uint8_t*)d_msgs32, (const uint8_t*)d_sigs96, n, dflags, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev[1], st));
    uint32_t fl[4];
    HIPCHK(hipMemcpyAsync(fl, c->d_flags, 16, hipMemcpyDeviceToHost, st));
    if (out_sets) HIPCHK(hipMemcpyAsync(out_sets, c->d_sets, n * 320, hipMemcpyDeviceToHost, st));
    if (status) HIPCHK(hipMemcpyAsync(status, c->d_status, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    HIPCHK(hipEventElapsedTime(&c->timings[0], c->ev[0], c->ev[1]));
    c->timings[7] = c->timings[0];
    return fl[2] ? 0 : 1;
}

// host wire-format arrays -> d_comp: keys at 0, messages at cap * 96, signatures at cap * 128
static int stage_compressed(mi355_bls_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint8_t* sigs, size_t n, uint32_t dflags) {
    {
        int rcr = io_reserve(c, n);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_comp, pks, n * ((dflags & DESER_F_PK_UNCOMPRESSED) ? 96 : 48), hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(c->d_comp + c->cap_io * 96, msgs, n * 32, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(c->d_comp + c->cap_io * 128, sigs, n * ((dflags & DESER_F_SIG_UNCOMPRESSED) ? 192 : 96), hipMemcpyHostToDevice, nullptr));
    return 0;
}
extern "C" int mi355_bls_deserialize_sets_device(mi355_bls_ctx* c, const void* d_pks48, const void* d_msgs32, const void* d_sigs96, size_t n, void* stream,
                                                 void* out_sets, uint8_t* status) {
    return mi355_bls_deserialize_sets_ex_device(c, d_pks48, d_msgs32, d_sigs96, n, 0, stream, out_sets, status);
}
extern "C" int mi355_bls_deserialize_sets_ex(mi355_bls_ctx* c, const uint8_t* pks, const uint8_t* msgs32, const uint8_t* sigs, size_t n, uint32_t dflags,
                                             void* out_sets, uint8_t* status) {
    if (!c || dflags > 7) return MI355_BLS_ERR_ARG;
    if (n == 0) return 1;
    if (!pks || !msgs32 || !sigs) return MI355_BLS_ERR_ARG;
    int rc = stage_compressed(c, pks, msgs32, sigs, n, dflags);
    if (rc) return rc;
    return mi355_bls_deserialize_sets_ex_device(c, c->d_comp, c->d_comp + c->cap_io * 96, c->d_comp + c->cap_io * 128, n, dflags, nullptr, out_sets, status);
}

extern "C" int mi355_bls_deserialize_sets(mi355_bls_ctx* c, const uint8_t* pks48, const uint8_t* msgs32, const uint8_t* sigs96, size_t n, void* out_sets,
                                          uint8_t* status) {
    if (!c) return MI355_BLS_ERR_ARG;
    if (n == 0) return 1;
    if (!pks48 || !msgs32 || !sigs96) return MI355_BLS_ERR_ARG;
    int rc = stage_compressed(c, pks48, msgs32, sigs96, n, 0);
    if (rc) return rc;
    return mi355_bls_deserialize_sets_device(c, c->d_comp, c->d_comp + c->cap_io * 96, c->d_comp + c->cap_io * 128, n, nullptr, out_sets, status);
}

extern "C" int mi355_bls_batch_verify_compressed_device(mi355_bls_ctx* c, const void* d_pks48, const void* d_msgs32, const void* d_sigs96, size_t n,
                                                        const uint8_t rnd[32], void* stream, uint8_t* status) {
    if (!c || !rnd) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!d_pks48 || !d_msgs32 || !d_sigs96) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev_deser0, st));
    int rc = deser_enqueue(c, (const uint8_t*)d_pks48, (const uint8_t*)d_msgs32, (const uint8_t*)d_sigs96, n, 0, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_deser1, st));
    uint32_t fl[4];
    HIPCHK(hipMemcpyAsync(fl, c->d_flags, 16, hipMemcpyDeviceToHost, st));
    if (status) HIPCHK(hipMemcpyAsync(status, c->d_status, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&c->deser_ms, c->ev_deser0, c->ev_deser1));
    if (fl[2]) return 0;                                   // some fromBytes failed: the caller never gets to batchVerify
    return verify_common(c, c->d_sets, nullptr, n, rnd, 0, st);
}

extern "C" int mi355_bls_batch_verify_compressed(mi355_bls_ctx* c, const uint8_t* pks48, const uint8_t* msgs32, const uint8_t* sigs96, size_t n,
                                                 const uint8_t rnd[32], uint8_t* status) {
    if (!c || !rnd) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;
    if (!pks48 || !msgs32 || !sigs96) return MI355_BLS_ERR_ARG;
    int rc = stage_compressed(c, pks48, msgs32, sigs96, n, 0);
    if (rc) return rc;
    return mi355_bls_batch_verify_compressed_device(c, c->d_comp, c->d_comp + c->cap_io * 96, c->d_comp + c->cap_io * 128, n, rnd, nullptr, status);
}

extern "C" float mi355_bls_last_deser_ms(mi355_bls_ctx* c) { return c ? c->deser_ms : 0.f; }

// ------------------------------------------------------------------------------------------
// Batch signer (test / bench input generation)
// ------------------------------------------------------------------------------------------
extern "C" int mi355_bls_sign_sets_device(mi355_bls_ctx* c, const void* d_sks32, const void* d_msgs32, size_t n, void* d_out_sets, void* stream,
                                          uint8_t* status) {
    if (!c) return MI355_BLS_ERR_ARG;
    if (n == 0) return 1;
    if (!d_sks32 || !d_msgs32 || !d_out_sets) return MI355_BLS_ERR_ARG;
    {
        int rcr = io_reserve(c, n);
        if (rcr) return rcr;
    }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
    HIPCHK(hipEventRecord(c->ev[0], st));
    uint32_t nb = ((uint32_t)n + WAVE - 1) / WAVE;
    k_sign_pk<<<nb, WAVE, 0, st>>>((const uint8_t*)d_sks32, (const uint8_t*)d_msgs32, (uint32_t)n, (uint8_t*)d_out_sets, c->d_status, c->d_flags);
    k_sign_sig<<<nb, WAVE, 0, st>>>((const uint8_t*)d_sks32, (const uint8_t*)d_msgs32, (uint32_t)n, c->dst, (uint8_t*)d_out_sets);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], st));
    uint32_t fl[4];
    HIPCHK(hipMemcpyAsync(fl, c->d_flags, 16, hipMemcpyDeviceToHost, st));
    if (status) HIPCHK(hipMemcpyAsync(status, c->d_status, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    HIPCHK(hipEventElapsedTime(&c->timings[0], c->ev[0], c->ev[1]));
    c->timings[7] = c->timings[0];
    return fl[2] ? 0 : 1;
}

extern "C" int mi355_bls_sign_sets(mi355_bls_ctx* c, const uint8_t* sks32, const uint8_t* msgs32, size_t n, void* out_sets, uint8_t* status) {
    if (!c) return MI355_BLS_ERR_ARG;
    if (n == 0) return 1;
    if (!sks32 || !msgs32 || !out_sets) return MI355_BLS_ERR_ARG;
    {
        int rcr = io_reserve(c, n);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_comp, sks32, n * 32, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(c->d_comp + c->cap_io * 48, msgs32, n * 32, hipMemcpyHostToDevice, nullptr));
    int rc = mi355_bls_sign_sets_device(c, c->d_comp, c->d_comp + c->cap_io * 48, n, c->d_sets, nullptr, status);
    if (rc < 0) return rc;
    HIPCHK(hipMemcpy(out_sets, c->d_sets, n * 320, hipMemcpyDeviceToHost));
    HIPCHK(hipMemsetAsync(c->d_comp, 0, n * 32, nullptr));          // do not leave the scalars in the staging buffer
    HIPCHK(hipStreamSynchronize(nullptr));
    return rc;
}

// ------------------------------------------------------------------------------------------
// combine
// ------------------------------------------------------------------------------------------
extern "C" int mi355_bls_combine(mi355_bls_ctx* c, const uint8_t rnd[32], const void* pks, const void* sigs, size_t n, uint8_t out_pk[96], uint8_t out_sig[192]) {
    if (!c || !rnd || !pks || !sigs || !out_pk || !out_sig || n == 0) return MI355_BLS_ERR_ARG;     // n == 0: the reference raises (core :584)
    if (n == 1) {                                                                                      // passthrough, no scalars (core :585-586)
        memcpy(out_pk, pks, 96);
        memcpy(out_sig, sigs, 192);
        return 0;
    }
    {
        int rcr = io_reserve(c, n);
        if (rcr) return rcr;
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = nullptr;
    uint8_t* d_pk = c->d_sets;                    // staging: n x 96 then n x 192 (<= n x 320)
    uint8_t* d_sg = c->d_sets + n * 96;
    HIPCHK(hipMemcpyAsync(d_pk, pks, n * 96, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_sg, sigs, n * 192, hipMemcpyHostToDevice, st));
    c->h_r.resize(n);
    host_combine_chain(rnd, n, c->h_r.data());
    HIPCHK(hipMemcpyAsync(c->d_r, c->h_r.data(), n * 8, hipMemcpyHostToDevice, st));
    // the reference's two 64-bit Pippenger calls (core :629-646): 8-byte scalars, nbits = 64
    // The two runs are independent: G1 on the caller's stream, G2 on the context's side stream with a workspace of its own, results
    // left on the device for `finish`; one synchronisation at the end (two blocking calls in a row: 7.2 ms at n = 4096).
    hipStream_t s2 = c->side ? c->side : st;
    HIPCHK(hipEventRecord(c->ev[0], st));
    if (s2 != st) HIPCHK(hipStreamWaitEvent(s2, c->ev[0], 0));            // the staged inputs
    int rc = msm_enqueue<fp2>(c, c->msm2, d_sg, n, c->d_r, 8, 64, s2, false, false);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev[2], s2));
    rc = msm_enqueue<fp>(c, c->msm, d_pk, n, c->d_r, 8, 64, st, false, s2 == st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev[1], st));
    if (s2 != st) HIPCHK(hipStreamWaitEvent(st, c->ev[2], 0));
    // `finish` (to affine, core :172-177)
    uint32_t* d_out = reinterpret_cast<uint32_t*>(c->d_msg);
    k_finish_affine<<<1, 1, 0, st>>>(c->msm->out, c->msm2->out, d_out, d_out + 24);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[3], st));
    HIPCHK(hipMemcpyAsync(out_pk, d_out, 96, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_sig, d_out + 24, 192, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 8; i++) c->timings[i] = 0;
    HIPCHK(hipEventElapsedTime(&c->timings[1], c->ev[0], c->ev[1]));       // G1 Pippenger
    HIPCHK(hipEventElapsedTime(&c->timings[2], c->ev[0], c->ev[2]));       // G2 Pippenger (beside it)
    HIPCHK(hipEventElapsedTime(&c->timings[7], c->ev[0], c->ev[3]));
    c->last_n = n;                  // fetch_stage(0) returns the combine scalars
    return 0;
}

// ------------------------------------------------------------------------------------------
// aggregateVerify
// ------------------------------------------------------------------------------------------
// One slice: pairs [0, n) of the slice (keys, rebased offsets and messages staged in d_sets), `with_sig`: the (-G1, sig) pair rides in
// this slice.  Leaves the slice's committed state in d_states slot 0 (final: also runs the final exponentiation, one-slice calls).
static int aggv_slice(mi355_bls_ctx* c, const uint8_t* pks, const uint8_t* msgs, const uint32_t* offs, size_t n, bool with_sig, bool final, hipStream_t st) {
    size_t total = offs[n];
    uint8_t* d_pk = c->d_sets;
    uint32_t* d_off = reinterpret_cast<uint32_t*>(c->d_sets + n * 96);
    uint8_t* d_msgs = c->d_sets + n * 96 + (n + 1) * 4;
    HIPCHK(hipMemcpyAsync(d_pk, pks, n * 96, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, offs, (n + 1) * 4, hipMemcpyHostToDevice, st));
    if (total) HIPCHK(hipMemcpyAsync(d_msgs, msgs, total, hipMemcpyHostToDevice, st));
    uint32_t n32 = (uint32_t)n, nb = (n32 + WAVE - 1) / WAVE, npairs = n32 + (with_sig ? 1u : 0u), nb1 = (npairs + WAVE - 1) / WAVE;
    HIPCHK(hipEventRecord(c->ev[0], st));
    bool all32 = c->xmd.valid;                      // every message 32 bytes long (signing roots): the batch path's hashing kernels
    for (size_t i = 0; i < n && all32; i++) all32 = offs[i + 1] - offs[i] == 32;
    if (all32) {
        // k_hash_map reads the message at offset 96 of a 320-byte record: the 32-byte messages are spread to that layout on the device
        // side of the staging buffer (keys | offsets | messages are packed at its start; the records go to d_comp)
        k_aggv_records<<<nb, WAVE, 0, st>>>(d_msgs, n32, c->d_comp);
        k_hash_map<<<(2 * n32 + WAVE - 1) / WAVE, WAVE, 0, st>>>(c->d_comp, n32, c->dst, c->xmd, c->d_M, c->mstride);
        if (c->coop && (n32 + 3) / 4 <= c->slots / 2)            // 16 lanes per message while that leaves half the wave slots free (at 4 096 messages it fills the chip and gains nothing)
            k_hash_clear_coop<16><<<(n32 + 3) / 4, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
        else if (c->coop && (n32 + 7) / 8 <= c->slots)
            k_hash_clear_coop<8><<<(n32 + 7) / 8, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
        else
            k_hash_clear<<<nb, WAVE, 0, st>>>(c->d_M, c->mstride, n32, c->d_H, c->stride);
    } else {
        k_hash_var<<<nb, WAVE, 0, st>>>(d_msgs, d_off, n32, c->dst, c->d_H, c->stride);
    }
    HIPCHK(hipEventRecord(c->ev[1], st));
    k_aggv_setup<<<nb1, WAVE, 0, st>>>(d_pk, n32, with_sig ? 1 : 0, reinterpret_cast<const uint32_t*>(c->d_msg + 4096), c->d_H, c->d_P, c->stride, c->d_flags);
    HIPCHK(hipEventRecord(c->ev[2], st));
    launch_lines(c, npairs, 0, st);
    HIPCHK(hipEventRecord(c->ev[3], st));
    {
        int rcp = enqueue_line_products(c, npairs, st, nullptr);
        if (rcp) return rcp;
    }
    HIPCHK(hipEventRecord(c->ev[4], st));
    k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, final ? 3 : 1, c->d_gt, c->d_flags + 1, 144, 0);
    HIPCHK(hipEventRecord(c->ev[5], st));
    HIPCHK(hipGetLastError());
    return 0;
}

// Any number of pairs: more than the context's capacity (pairs, or staged bytes: keys + offsets + messages share d_sets) are
// processed in slices whose committed states are multiplied on the engine (k_state_mul), as for batchVerify.
static int aggregate_verify_impl(mi355_bls_ctx* c, const void* pks, const uint8_t* msgs, const uint32_t* msg_offsets, size_t n, const void* sig, bool sig_is_p2) {
    if (!c || !sig) return MI355_BLS_ERR_ARG;
    if (n == 0) return 0;                                   // "Spec precondition" (bls_sig_min_pubkey.nim:165-167)
    if (!pks || !msgs || !msg_offsets) return MI355_BLS_ERR_ARG;
    for (size_t i = 0; i < n; i++)
        if (msg_offsets[i] > msg_offsets[i + 1]) return MI355_BLS_ERR_ARG;      // offsets must be non-decreasing (lengths are differences)
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = nullptr;
    HIPCHK(hipMemsetAsync(c->d_flags, 0, 12, st));
    if (sig_is_p2) {                                        // AggregateSignature (blst_p2, Jacobian): to affine on the device, as finish() does
        HIPCHK(hipMemcpyAsync(c->d_msg + 4096 + 256, sig, 288, hipMemcpyHostToDevice, st));
        k_p2_to_affine<<<1, 1, 0, st>>>(reinterpret_cast<const uint32_t*>(c->d_msg + 4096 + 256), reinterpret_cast<uint32_t*>(c->d_msg + 4096));
    } else {
        HIPCHK(hipMemcpyAsync(c->d_msg + 4096, sig, 192, hipMemcpyHostToDevice, st));
    }
    const size_t budget = c->cap * 320;
    std::vector<uint32_t> offs;
    size_t a = 0;
    uint32_t slice = 0;
    bool single = true;
    while (a < n) {
        // greedy slice [a, b): at most cap pairs, staged bytes within d_sets
        size_t b = a, bytes = 4;
        while (b < n && b - a < c->cap) {
            size_t add = 96 + 4 + (msg_offsets[b + 1] - msg_offsets[b]);
            if (bytes + add > budget) break;
            bytes += add;
            b++;
        }
        if (b == a) {
            g_err = "one message does not fit the context's staging buffer";
            return MI355_BLS_ERR_CAPACITY;
        }
        const bool last = b == n;
        if (slice == 0) single = last;
        offs.resize(b - a + 1);
        for (size_t i = a; i <= b; i++) offs[i - a] = msg_offsets[i] - msg_offsets[a];
        int rc = aggv_slice(c, (const uint8_t*)pks + 96 * a, msgs + msg_offsets[a], offs.data(), b - a, last, single, st);
        if (rc) return rc;
        if (!single) {
            k_state_mul<<<1, TAIL_THREADS, 0, st>>>(c->d_states, 1, slice ? 1 : 0, slice ? 0 : -1);
            HIPCHK(hipStreamSynchronize(st));              // offs (host vector) and the staging buffer are reused by the next slice
        }
        a = b;
        slice++;
    }
    if (!single) {
        k_state_mul<<<1, TAIL_THREADS, 0, st>>>(c->d_states, 0, 1, -1);
        k_tail<<<1, tail_threads(c), 0, st>>>(c->d_L, c->d_states, 1, 2, c->d_gt, c->d_flags + 1, 144, 0);
        HIPCHK(hipEventRecord(c->ev[5], st));
        HIPCHK(hipGetLastError());
    }
    uint32_t fl[2];
    HIPCHK(hipMemcpyAsync(fl, c->d_flags, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    c->have_gt = true;
    c->gt_is_fv = false;
    c->last_n = 0;
    int rc = collect_timings(c, 5);
    if (rc) return rc;
    return (fl[0] == 0 && fl[1] == 1) ? 1 : 0;
}

extern "C" int mi355_bls_aggregate_verify(mi355_bls_ctx* c, const void* pks, const uint8_t* msgs, const uint32_t* msg_offsets, size_t n, const void* sig) {
    return aggregate_verify_impl(c, pks, msgs, msg_offsets, n, sig, false);
}
// the same with the signature as an AggregateSignature (blst_p2, Jacobian, 288 B): finish(AggregateSignature), core :357
extern "C" int mi355_bls_aggregate_verify_p2(mi355_bls_ctx* c, const void* pks, const uint8_t* msgs, const uint32_t* msg_offsets, size_t n, const void* sig_p2) {
    return aggregate_verify_impl(c, pks, msgs, msg_offsets, n, sig_p2, true);
}

// ContextCoreAggregateVerify (blst_min_pubkey_sig_core.nim:305-414), the streaming form: init / update(publicKey, message) /
// finish(signature).  The pairs are collected on the host (176 bytes + the message each) and verified by ONE device call at
// finish - the reference's update also only queues work that commit / finalVerify later complete (blst's N_MAX = 8 pair buffer).
// update returns 0 for the infinity public key (BLST_PK_IS_INFINITY: the reference's update returns false) and the context
// stays failed until the next init.
extern "C" int mi355_bls_aggv_init(mi355_bls_ctx* c) {
    if (!c) return MI355_BLS_ERR_ARG;
    c->av_pks.clear();
    c->av_msgs.clear();
    c->av_offs.assign(1, 0u);
    c->av_failed = false;
    return 0;
}
extern "C" int mi355_bls_aggv_update(mi355_bls_ctx* c, const void* pk, const uint8_t* msg, size_t msg_len) {
    if (!c || !pk || (!msg && msg_len) || c->av_offs.empty() || msg_len > (1u << 30)) return MI355_BLS_ERR_ARG;
    const uint8_t* p = (const uint8_t*)pk;
    bool inf = true;
    for (int i = 0; i < 96; i++) inf = inf && p[i] == 0;
    if (inf) {
        c->av_failed = true;
        return 0;
    }
    if (c->av_msgs.size() + msg_len > 0xffffffffull) {   // message offsets are 32-bit: refuse instead of wrapping (the context stays usable)
        g_err = "mi355_bls_aggv_update: more than 4 GiB of messages in one aggregateVerify";
        return MI355_BLS_ERR_CAPACITY;
    }
    c->av_pks.insert(c->av_pks.end(), p, p + 96);
    if (msg_len) c->av_msgs.insert(c->av_msgs.end(), msg, msg + msg_len);
    c->av_offs.push_back((uint32_t)c->av_msgs.size());
    return 1;
}
static int aggv_finish_impl(mi355_bls_ctx* c, const void* sig, bool sig_is_p2) {
    if (!c || !sig || c->av_offs.empty()) return MI355_BLS_ERR_ARG;
    size_t n = c->av_offs.size() - 1;
    int rc = 0;
    // no pair seen: blst's finalverify has no GT accumulator set -> false; a failed update -> false
    if (!c->av_failed && n) {
        uint8_t dummy = 0;
        rc = aggregate_verify_impl(c, c->av_pks.data(), c->av_msgs.empty() ? &dummy : c->av_msgs.data(), c->av_offs.data(), n, sig, sig_is_p2);
    }
    c->av_offs.clear();                                  // finish consumes the context: init again before the next use
    c->av_pks.clear();
    c->av_msgs.clear();
    return rc;
}
extern "C" int mi355_bls_aggv_finish(mi355_bls_ctx* c, const void* sig) { return aggv_finish_impl(c, sig, false); }
// finish(signature: AggregateSignature) (blst_min_pubkey_sig_core.nim:357): the Jacobian blst_p2 image, 288 B
extern "C" int mi355_bls_aggv_finish_p2(mi355_bls_ctx* c, const void* sig_p2) { return aggv_finish_impl(c, sig_p2, true); }

// ------------------------------------------------------------------------------------------
// Point-sharded MSM across devices ("MSM"): every device computes the full-width partial sum of its
// share of the points (blst_p1 / blst_p2, Jacobian), the partials are added (blst_p1_add_or_double, blst_abi.nim:278) - 144 or
// 288 bytes per device is all that is exchanged.
// ------------------------------------------------------------------------------------------
template <class F>
static int jac_sum_device(mi355_bls_ctx* c, uint8_t* ret, const void* d_parts, size_t k, size_t stride_bytes, hipStream_t st) {
    constexpr size_t JACB = (sizeof(F) == sizeof(fp) ? 96 : 192) / 2 * 3;
    if (!c || !ret || !d_parts || k == 0 || k > 4096 || stride_bytes < JACB || (stride_bytes & 3)) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    k_jac_sum_blst<F><<<1, WAVE, 0, st>>>((const uint32_t*)d_parts, (uint32_t)k, (uint32_t)(stride_bytes / 4), c->d_agg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ret, c->d_agg, JACB, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    c->agg_valid = false;                                // d_agg doubles as the batch path's aggregate-signature buffer
    return 0;
}
template <class F>
static int jac_sum_host(mi355_bls_ctx* c, uint8_t* ret, const uint8_t* parts, size_t k) {
    constexpr size_t JACB = (sizeof(F) == sizeof(fp) ? 96 : 192) / 2 * 3;
    if (!c || !ret || !parts || k == 0 || k * JACB > 2048 * 2 * G1W * 4) return MI355_BLS_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_export, parts, k * JACB, hipMemcpyHostToDevice, nullptr));
    return jac_sum_device<F>(c, ret, c->d_export, k, JACB, nullptr);
}
extern "C" int mi355_bls_p1s_add(mi355_bls_ctx* c, uint8_t ret_p1[144], const uint8_t* parts, size_t k) { return jac_sum_host<fp>(c, ret_p1, parts, k); }
extern "C" int mi355_bls_p2s_add(mi355_bls_ctx* c, uint8_t ret_p2[288], const uint8_t* parts, size_t k) { return jac_sum_host<fp2>(c, ret_p2, parts, k); }
extern "C" int mi355_bls_p1s_add_device(mi355_bls_ctx* c, uint8_t ret_p1[144], const void* d_parts, size_t k, size_t stride_bytes, void* stream) {
    return jac_sum_device<fp>(c, ret_p1, d_parts, k, stride_bytes, (hipStream_t)stream);
}
// this device's partial of a point-sharded MSM, left in DEVICE memory (d_out_p1, 144 B) behind everything else on `stream`: the
// send buffer of the collective that gathers the partials; nothing is waited for
extern "C" int mi355_bls_p1s_mult_pippenger_partial_device(mi355_bls_ctx* c, void* d_out_p1, const void* d_points, size_t npoints, const void* d_scalars,
                                                           size_t nbits, void* stream) {
    if (!c || !d_out_p1 || nbits == 0 || nbits > 256 || npoints > (1u << 28)) return MI355_BLS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    if (npoints == 0) {
        HIPCHK(hipMemsetAsync(d_out_p1, 0, 144, st));
        return 0;
    }
    if (!d_points || !d_scalars) return MI355_BLS_ERR_ARG;
    int rc = msm_enqueue<fp>(c, c->msm, d_points, npoints, d_scalars, 32, nbits, st, false, true);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d_out_p1, c->msm->out, 144, hipMemcpyDeviceToDevice, st));
    return 0;
}

// One host thread, ngpu contexts (one per device): device g takes points [off_g, off_g + cnt_g) (balanced contiguous blocks), all
// partial MSMs are enqueued before any is waited for, the partials return through pinned host memory and ctxs[0] adds them.
// Host arrays (pts / sc) or per-device resident arrays (d_pts[g] / d_sc[g] hold shard g).
template <class F>
static int msm_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t* ret, const uint8_t* pts, const uint8_t* sc, const void* const d_pts[],
                     const void* const d_sc[], size_t npoints, size_t nbits) {
    constexpr size_t AFFB = sizeof(F) == sizeof(fp) ? 96 : 192, JACB = AFFB / 2 * 3;
    if (!ctxs || ngpu == 0 || ngpu > 64 || !ret || nbits == 0 || nbits > 256 || npoints > (1u << 28)) return MI355_BLS_ERR_ARG;
    if (npoints == 0) {
        std::memset(ret, 0, JACB);
        return 0;
    }
    for (size_t g = 0; g < ngpu; g++)
        if (!ctxs[g]) return MI355_BLS_ERR_ARG;
    if (!(pts && sc) && !(d_pts && d_sc)) return MI355_BLS_ERR_ARG;
    size_t base = npoints / ngpu, rem = npoints % ngpu;
    bool reg_p = false, reg_s = false;
    if (pts) {                                           // page-lock the caller's arrays: every device's copy is then a real asynchronous DMA
        reg_p = hipHostRegister(const_cast<uint8_t*>(pts), npoints * AFFB, hipHostRegisterPortable) == hipSuccess;
        reg_s = hipHostRegister(const_cast<uint8_t*>(sc), npoints * 32, hipHostRegisterPortable) == hipSuccess;
        (void)hipGetLastError();
    }
    bool live[64] = {};
    int rc = 0;
    for (size_t g = 0; g < ngpu && !rc; g++) {
        size_t off = g < rem ? (base + 1) * g : base * g + rem, cnt = base + (g < rem ? 1 : 0);
        if (cnt == 0) continue;
        mi355_bls_ctx* c = ctxs[g];
        if (hipSetDevice(c->device) != hipSuccess) { g_err = "hipSetDevice"; rc = MI355_BLS_ERR_HIP; break; }
        const void *dp = d_pts ? d_pts[g] : nullptr, *ds = d_sc ? d_sc[g] : nullptr;
        if (!dp || !ds) {
            if (!pts || !sc) { rc = MI355_BLS_ERR_ARG; break; }
            rc = msm_reserve(c, c->msm, cnt, pip_plan(cnt, nbits), AFFB);
            if (rc) break;
            if (hipMemcpyAsync(c->msm->d_pts, pts + off * AFFB, cnt * AFFB, hipMemcpyHostToDevice, nullptr) != hipSuccess ||
                hipMemcpyAsync(c->msm->d_sc, sc + off * 32, cnt * 32, hipMemcpyHostToDevice, nullptr) != hipSuccess) {
                g_err = "hipMemcpyAsync (MSM shard staging)";
                rc = MI355_BLS_ERR_HIP;
                break;
            }
            dp = c->msm->d_pts;
            ds = c->msm->d_sc;
        }
        rc = msm_enqueue<F>(c, c->msm, dp, cnt, ds, 32, nbits, nullptr, false, true);
        if (rc) break;
        if (hipMemcpyAsync(c->h_flags + 160, c->msm->out, JACB, hipMemcpyDeviceToHost, nullptr) != hipSuccess) { g_err = "hipMemcpyAsync (MSM partial)"; rc = MI355_BLS_ERR_HIP; break; }
        live[g] = true;
    }
    std::vector<uint8_t> parts;
    for (size_t g = 0; g < ngpu; g++) {                  // every device that was handed work is waited for, also after a failure
        if (!live[g]) continue;
        (void)hipSetDevice(ctxs[g]->device);
        if (hipStreamSynchronize(nullptr) != hipSuccess && !rc) { g_err = "hipStreamSynchronize (MSM shard)"; rc = MI355_BLS_ERR_HIP; }
        const uint8_t* h = reinterpret_cast<const uint8_t*>(ctxs[g]->h_flags + 160);
        parts.insert(parts.end(), h, h + JACB);
    }
    if (reg_p) (void)hipHostUnregister(const_cast<uint8_t*>(pts));
    if (reg_s) (void)hipHostUnregister(const_cast<uint8_t*>(sc));
    if (rc) return rc;
    return jac_sum_host<F>(ctxs[0], ret, parts.data(), parts.size() / JACB);
}
extern "C" int mi355_bls_p1s_mult_pippenger_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p1[144], const void* const points[], size_t npoints,
                                                  const uint8_t* const scalars[], size_t nbits) {
    if (npoints && (!points || !points[0] || !scalars || !scalars[0])) return MI355_BLS_ERR_ARG;
    return msm_multi<fp>(ctxs, ngpu, ret_p1, npoints ? (const uint8_t*)points[0] : nullptr, npoints ? scalars[0] : nullptr, nullptr, nullptr, npoints, nbits);
}
extern "C" int mi355_bls_p2s_mult_pippenger_multi(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p2[288], const void* const points[], size_t npoints,
                                                  const uint8_t* const scalars[], size_t nbits) {
    if (npoints && (!points || !points[0] || !scalars || !scalars[0])) return MI355_BLS_ERR_ARG;
    return msm_multi<fp2>(ctxs, ngpu, ret_p2, npoints ? (const uint8_t*)points[0] : nullptr, npoints ? scalars[0] : nullptr, nullptr, nullptr, npoints, nbits);
}
extern "C" int mi355_bls_p1s_mult_pippenger_multi_device(mi355_bls_ctx* const ctxs[], size_t ngpu, uint8_t ret_p1[144], const void* const d_points[], size_t npoints,
                                                         const void* const d_scalars[], size_t nbits) {
    if (npoints && (!d_points || !d_scalars)) return MI355_BLS_ERR_ARG;
    return msm_multi<fp>(ctxs, ngpu, ret_p1, nullptr, nullptr, d_points, d_scalars, npoints, nbits);
}
/* how mi355_bls_p1s_mult_pippenger_multi cuts npoints into ngpu contiguous shards */
extern "C" void mi355_bls_msm_shard_range(size_t npoints, uint32_t world, uint32_t rank, size_t* first, size_t* count) {
    size_t base = world ? npoints / world : 0, rem = world ? npoints % world : 0;
    *first = rank < rem ? (base + 1) * rank : base * rank + rem;
    *count = base + (rank < rem ? 1 : 0);
}
