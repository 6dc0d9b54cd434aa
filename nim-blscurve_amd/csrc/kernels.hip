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
#include <atomic>
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
#include "teamvm.hpp"
#include "rowfp.hpp"
#include "rowvm.hpp"

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

// Experiment knobs of the measurement scripts under tools/ (MI355_BLS_CHAIN_EV, MI355_BLS_MSM_SEG / _TEAM / _CUTS) exist only in builds
// made with -DBLS_EXPERIMENTS; the product reads two environment variables, both documented in the header: MI355_BLS_NO_ENV and
// MI355_BLS_DEVICE.
#ifdef BLS_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

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
// t: the (message, u) pair this lane maps; store: whether it writes the result (the row form below computes every pair in sixteen lanes)
template <class Pow>
__device__ __forceinline__ void hash_map_body_with(const uint8_t* __restrict__ sets, uint32_t t, bool store, const dst_t& dst, const xmd32_consts& xc, uint4* __restrict__ M,
                                                   size_t mstride, const Pow& pw) {
    const uint32_t i = t >> 1;
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
    const g2_jac q = iso3_g2(sswu_g2_with(u, pw));
    if (store) soa_st_g2(M, mstride, t, q);
}
__device__ __forceinline__ void hash_map_body(const uint8_t* __restrict__ sets, uint32_t n, const dst_t& dst, const xmd32_consts& xc, uint4* __restrict__ M, size_t mstride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((t >> 1) >= n) return;
    hash_map_body_with(sets, t, true, dst, xc, M, mstride, pow_in_lane{});
}
__global__ void __launch_bounds__(WAVE, 2) k_hash_map(const uint8_t* __restrict__ sets, uint32_t n, dst_t dst, xmd32_consts xc, uint4* __restrict__ M,
                                                         size_t mstride) {
    hash_map_body(sets, n, dst, xc, M, mstride);
}
// The same kernel for grids of at most one wave per SIMD (latency mode, up to 32 768 messages): the whole register file (nothing spills around the
// square-root chains) and one wave per SIMD guaranteed - the dispatcher packs the 256-register form two per SIMD before every SIMD has a wave, and a
// wave that shares its SIMD takes 1.4 x as long.
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_hash_map_spread(const uint8_t* __restrict__ sets, uint32_t n, dst_t dst, xmd32_consts xc, uint4* __restrict__ M, size_t mstride) {
    hash_map_body(sets, n, dst, xc, M, mstride);
}
// Small batches (at most one wave per SIMD at FOUR pairs per wave): a (message, u) pair per DPP row - the sixteen lanes of a row run the same hashing and
// map arithmetic, and the two square-root exponentiations, 924 dependent products that are most of this kernel's time, run ALONG the row (rowfp.hpp:
// ~0.44 us per product where a lane takes 0.7 - 0.9).
#if defined(__HIP_DEVICE_COMPILE__)
template <int I>
__device__ __forceinline__ void row_spread_limbs(rw r, fp& out) {            // limb I of the row's value into every lane of the row
    out.l[I] = (uint32_t)row_bcast<I>(r);
    if constexpr (I + 1 < FP_N) row_spread_limbs<I + 1>(r, out);
}
struct pow_per_row {
    const row_ctx& C;
    uint32_t* tab;                    // LDS: 16 entries x 64 lanes, this lane's column
    __device__ __forceinline__ fp operator()(const fp& a) const {          // a: the same in the sixteen lanes of a row
        row_tab_lds T{tab, (uint32_t)WAVE};
        const rw r = row_pow_sched(C, row_from_fp(a), k::SW_PM3D4, k::SW_PM3D4_LEN, T);
        fp out;
        row_spread_limbs<0>(r, out);
        return fp_reduce(out);
    }
};
#endif
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_hash_map_rows(const uint8_t* __restrict__ sets, uint32_t n, dst_t dst, xmd32_consts xc, uint4* __restrict__ M, size_t mstride) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t powtab[16 * WAVE];
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool live = t < 2 * n;
    const row_ctx RC = row_ctx_make();
    hash_map_body_with(sets, live ? t : 0u, live && (threadIdx.x & 15u) == 0, dst, xc, M, mstride, pow_per_row{RC, powtab + threadIdx.x});
#endif
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
// Round 5: the WHOLE kernel body is one hand-allocated assembly statement (tools/gen_clear_asm.py -> build/clear_asm.inc; curve.hpp's
// jac_dbl_lazy / jac_precompute + jac_add_pre / g2_psi in h2c.hpp's clear_cofactor_g2_chain order, executed and checked end to end by
// tests/test_asm_loops.py): loads of the two mapped points, P = q0 + q1, both doubling chains, the psi maps, the seven additions around them,
// the store.  (Moving only the chains into assembly gained nothing - the compiled chain already ran at 4.2 cycles per instruction; the seven
// out-of-line complete additions around it ran at 7.3, their 252 argument words each travelling through scratch memory: 11 % of the
// instructions, 17 % of the time.)  Points that outlive a chain wait in three LDS slots and in two columns per lane of `scratch` (the
// context's line store, unused until k_lines; other streams only touch columns >= n of it).  A lane that met an exceptional case of the
// incomplete addition formulas (a Z that turned out zero: operand at infinity, P == +-Q) comes back flagged and is recomputed here with the
// complete compiled formulas - never taken for hash outputs, exercised by mi355_bls_debug_g2_clear_cofactor.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_CLEAR_NOASM)
#include "../build/clear_asm.inc"
__device__ __forceinline__ uint32_t clear_asm(const uint4* M, uint32_t mstride16, uint4* H, uint32_t stride16, uint4* scratch, uint32_t sstride16, uint32_t i, uint32_t lds) {
    uint32_t flag;
    asm volatile(BLS_CLEAR_ASM_BODY : "=v"(flag) : "s"(M), "s"(mstride16), "s"(H), "s"(stride16), "s"(scratch), "s"(sstride16), "v"(i), "s"(lds) : BLS_CLEAR_ASM_CLOBBERS);
    return flag;
}
#endif
__global__ void __launch_bounds__(WAVE) k_hash_clear(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride, uint4* __restrict__ scratch) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 park_slots[3 * BLS_LDS_SLOT];
    g2_park_lds park{(bls_lds_u32x4*)park_slots};
#else
    g2_park_regs park;               // host pass of the translation unit: kernels are parsed, never run
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_CLEAR_NOASM)
    if (scratch && mstride * 16 * 24 < ((size_t)1 << 32) && stride * 16 * 24 < ((size_t)1 << 32)) {      // wave-uniform: the loop's row arithmetic is 32-bit
        const uint32_t flag = clear_asm(M, (uint32_t)(mstride * 16), H, (uint32_t)(stride * 16), scratch, (uint32_t)(stride * 16), i, (uint32_t)(uintptr_t)park.base);
        if (flag) {                     // complete formulas for this lane (the out-of-line compact form)
            g2_jac q0 = soa_ld_g2(M, mstride, 2 * (size_t)i), q1 = soa_ld_g2(M, mstride, 2 * (size_t)i + 1);
            soa_st_g2(H, stride, i, clear_cofactor_g2(jac_add(q0, q1)));
        }
        return;
    }
#endif
    g2_jac q0 = soa_ld_g2(M, mstride, 2 * (size_t)i), q1 = soa_ld_g2(M, mstride, 2 * (size_t)i + 1);
    soa_st_g2(H, stride, i, clear_cofactor_g2_with(jac_add(q0, q1), park, mul_inplace{}));
}
#if defined(BLS_CLEAR_TWO_WAVE)
// EXPERIMENT (round 6, review item 4; builds made with -DBLS_CLEAR_TWO_WAVE only): the same generated kernel body for 256 registers and two
// waves per SIMD (tools/gen_clear_asm.py --two-wave): no AGPRs, no LDS - what the one-wave form parks there travels through a wave-private block of
// global memory (252 rows of 256 bytes) and a third scratch column; the doubling loop touches neither.  A flagged lane leaves Z = 0 for
// k_clear_fix (the complete formulas need the whole register file).  `scratch` (the context's line store): three per-lane columns, rows 0 .. 71;
// `wave_blocks`: k_pkmul's table buffer, idle while the hashing runs (the line store's columns >= n belong to the extra pairs' lines, which a
// latency-mode call writes on the fork stream at the same time).
#if defined(__HIP_DEVICE_COMPILE__)
#include "../build/clear2_asm.inc"
#endif
__global__ void __launch_bounds__(WAVE, 2) k_hash_clear2(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride, uint4* __restrict__ scratch,
                                                          uint4* __restrict__ wave_blocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* wblock = wave_blocks + (size_t)blockIdx.x * (252 * 256 / 16);
    uint32_t flag;
    asm volatile(BLS_CLEAR2_ASM_BODY
                 : "=v"(flag)
                 : "s"(M), "s"((uint32_t)(mstride * 16)), "s"(H), "s"((uint32_t)(stride * 16)), "s"(scratch), "s"((uint32_t)(stride * 16)), "v"(i), "s"(0u), "s"(wblock)
                 : BLS_CLEAR2_ASM_CLOBBERS);
    if (flag) {                                   // mark for k_clear_fix: Z = 0
        soa_st2(H, stride, 4, i, fp2_zero());
    }
#endif
}
#endif
// test entry (mi355_bls_debug_g2_clear_cofactor): pairs of blst_p2 images -> the SoA layout k_hash_map leaves its mapped points in
__global__ void __launch_bounds__(WAVE) k_debug_to_soa(const uint32_t* __restrict__ in, uint32_t npoints, uint4* __restrict__ M, size_t mstride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoints) return;
    soa_st_g2(M, mstride, i, ld_g2_blst(in + (size_t)i * 72));
}

// Device teams of the compiled lane-cooperative formulas (curve.hpp jac_dbl_team / jac_add_team): lanes gbase .. gbase + 7 hold the same values;
// roles 0.. take one product each of a round (ONE multiplier call with per-lane operands), then every lane reads all results with wave
// shuffles.  Rounds 1-5 ran the latency path's hashing and Miller lines this way (k_hash_clear_coop, k_lines_coop, 8 or 16 lanes per item:
// ~700 spilled registers, ~5 300 cycles per round); round 6 moved those to the lane-team engine below (csrc/teamvm.hpp).  What is left here
// serves the doubling chains of the G2 Pippenger's window sums (dbl_coop).
template <int CTRL>
__device__ __forceinline__ fp fp_quad_perm(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], CTRL, 0xf, 0xf, true);
    return r;
}
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
// ------------------------------------------------------------------------------------------
// Round 6: the same two jobs on the LANE-TEAM ENGINE (csrc/teamvm.hpp, programs from tools/teamvm.py): 16 lanes per message / pair, the
// team's values in LDS slots, every formula a table of rounds (one Fp product per lane and round, quad-local linear combination folded
// into the reduction).  No selects, no shuffles, nothing spilled.  Four teams per wave.
// ------------------------------------------------------------------------------------------
#define TVM_TABLE __device__
#include "../build/teamvm_tables.inc"
#if defined(__HIP_DEVICE_COMPILE__)
// the team's region of a wave's LDS block, and the lane's place in its team
template <uint32_t SLOTS>
__device__ __forceinline__ tvm_lds_char* tvm_team_base(bls_u32x4* lds) {
    return (tvm_lds_char*)(tvm_lds_u32x4*)lds + (threadIdx.x >> 4) * (SLOTS * TVM_SLOT_BYTES);
}
__device__ __forceinline__ bool tvm_slot_is_zero(const tvm_lds_char* team, uint32_t slot) {       // a slot holds |v| < 0.51 p: 0 mod p <=> every limb is 0
    return fp_limbs_are_zero(tvm_ld(team, slot * TVM_SLOT_BYTES));
}
// cofactor clearing of the team's two mapped points (already in the slots X.. and BX.., partially reduced): H in X, Y, Z.  Returns false when
// the result's Z is 0 - an exceptional case of the incomplete additions on the way (operand at infinity, P = +-Q: every later step keeps Z = 0)
// or a genuine point at infinity: the caller recomputes with the complete formulas.
__device__ __forceinline__ bool tvm_clear_cofactor(tvm_lds_char* team, uint32_t lane16) {
    if (lane16 == 0) tvm_st(team, TVM_CLEAR_zero * TVM_SLOT_BYTES, fp_zero());
    if (lane16 >= 12) {
        const fp2 cx = fp2_from_const(k::PSI_CX), cy = fp2_from_const(k::PSI_CY);
        tvm_st(team, (TVM_CLEAR_CX + (lane16 - 12)) * TVM_SLOT_BYTES, fp_reduce(fp_select(lane16 < 14, fp_select(lane16 == 12, cx.c0, cx.c1), fp_select(lane16 == 14, cy.c0, cy.c1))));
    }
    static_assert(TVM_CLEAR_CY == TVM_CLEAR_CX + 2 && TVM_CLEAR_BX == TVM_CLEAR_X + 6, "slot order the prologue relies on");
    tvm_run<false>(team, lane16, TVM_CLEAR_DESC, TVM_CLEAR_SEQ, TVM_CLEAR_NSEQ, tvm_line_sink{});
    return !(tvm_slot_is_zero(team, TVM_CLEAR_Z) & tvm_slot_is_zero(team, TVM_CLEAR_Z + 1));
}
#endif
// a^((p-3)/4) for the operands of lanes 0 and 1 of a wave, each along the rows of its parity; every lane gets the result of lane (lane & 15) == 1 ? 1 : 0
struct pow_two_rows {
    const row_ctx& C;
    uint32_t* tab;                    // LDS: 16 entries x 64 lanes, this lane's column
    __device__ __forceinline__ fp operator()(const fp& a) const {
        const uint32_t l16 = threadIdx.x & 15u;
        const bool odd = (threadIdx.x & 16u) != 0;
        rw x = 0;
#pragma unroll
        for (int i = 0; i < FP_N; i++) {
            const uint32_t v0 = __builtin_amdgcn_readlane(a.l[i], 0), v1 = __builtin_amdgcn_readlane(a.l[i], 1);
            x = l16 == (uint32_t)i ? (rw)(odd ? v1 : v0) : x;
        }
        row_tab_lds T{tab, (uint32_t)WAVE};
        const rw r = row_pow_sched(C, x, k::SW_PM3D4, k::SW_PM3D4_LEN, T);
        fp out;
#pragma unroll
        for (int i = 0; i < FP_N; i++) {
            const uint32_t v0 = __builtin_amdgcn_readlane(r, i), v1 = __builtin_amdgcn_readlane(r, 16 + i);
            out.l[i] = l16 == 1u ? v1 : v0;
        }
        return fp_reduce(out);
    }
};
// ONE message of any length (fastAggregateVerify / coreVerify shape): latency is all that matters, so a wave works on
// it cooperatively: the two SSWU maps run in roles 0 and 1, the doubling chains of the cofactor clearing spread
// their independent products over roles 0..2.  Every group of 8 lanes does the same work.
__global__ void __launch_bounds__(256) k_hash_one(const uint8_t* __restrict__ msg, uint32_t len, dst_t dst, xmd32_consts xc, uint4* __restrict__ H, size_t stride, size_t slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 256 lanes: wave 0 hashes and maps (the two SSWU maps in lanes 0 and 1 of every team of 16, their square-root chains along DPP rows); then the
    // cofactor clearing runs on the ROW executor (rowvm.hpp): the engine's 486 rounds with a row of 16 lanes per product, four waves for the one message
    __shared__ bls_u32x4 lds[TVM_CLEAR_SLOTS * (TVM_SLOT_BYTES / 16)];
    __shared__ uint32_t powtab[16 * WAVE];
    tvm_lds_char* item = (tvm_lds_char*)(tvm_lds_u32x4*)lds;
    const row_ctx RC = row_ctx_make();
    g2_jac q;
    if (threadIdx.x < WAVE) {
        const uint32_t lane16 = threadIdx.x & 15u;
        fp2 u0, u1;
        if (xc.valid && len == 32) {                      // the usual message (a 32-byte signing root): the batch path's 18 compressions with the DST's words prepared
            uint32_t mbe[8];                              // on the host - the byte-wise absorber below costs 0.23 ms for such a message, this form 0.05
#pragma unroll
            for (int j = 0; j < 8; j++) mbe[j] = ((uint32_t)msg[4 * j] << 24) | ((uint32_t)msg[4 * j + 1] << 16) | ((uint32_t)msg[4 * j + 2] << 8) | (uint32_t)msg[4 * j + 3];
            hash_to_field_fp2x2_msg32(u0, u1, mbe, xc);
        } else {
            hash_to_field_fp2x2(u0, u1, msg, len, dst.b, dst.len);
        }
        // rows 0 / 2 take lane 0's operand of an exponentiation, rows 1 / 3 lane 1's (pow_two_rows)
        q = iso3_g2(sswu_g2_with(fp2_select(lane16 == 1, u1, u0), pow_two_rows{RC, powtab + threadIdx.x}));
        if (threadIdx.x < 2) {                                          // q0 -> the engine's slots X, Y, Z; q1 -> BX, BY, BZ
            const uint32_t s0 = (TVM_CLEAR_X + 6 * threadIdx.x) * TVM_SLOT_BYTES;
            tvm_st(item, s0, fp_reduce(q.x.c0)); tvm_st(item, s0 + TVM_SLOT_BYTES, fp_reduce(q.x.c1));
            tvm_st(item, s0 + 2 * TVM_SLOT_BYTES, fp_reduce(q.y.c0)); tvm_st(item, s0 + 3 * TVM_SLOT_BYTES, fp_reduce(q.y.c1));
            tvm_st(item, s0 + 4 * TVM_SLOT_BYTES, fp_reduce(q.z.c0)); tvm_st(item, s0 + 5 * TVM_SLOT_BYTES, fp_reduce(q.z.c1));
        } else if (threadIdx.x == 2) {
            tvm_st(item, TVM_CLEAR_zero * TVM_SLOT_BYTES, fp_zero());
        } else if (threadIdx.x >= 12 && threadIdx.x < 16) {
            const fp2 cx = fp2_from_const(k::PSI_CX), cy = fp2_from_const(k::PSI_CY);
            tvm_st(item, (TVM_CLEAR_CX + (threadIdx.x - 12)) * TVM_SLOT_BYTES,
                   fp_reduce(fp_select(threadIdx.x < 14, fp_select(threadIdx.x == 12, cx.c0, cx.c1), fp_select(threadIdx.x == 14, cy.c0, cy.c1))));
        }
    }
    __syncthreads();
    rvm_run<false>(RC, item, TVM_CLEAR_DESC, TVM_CLEAR_SEQ, TVM_CLEAR_NSEQ, tvm_line_sink{});
    if (threadIdx.x >= WAVE) return;
    // Z = 0 mod p: an exceptional case of the incomplete additions on the way (or a true point at infinity) - the complete formulas on one lane
    const bool ok = !(fp_is_zero(tvm_ld(item, TVM_CLEAR_Z * TVM_SLOT_BYTES)) & fp_is_zero(tvm_ld(item, (TVM_CLEAR_Z + 1) * TVM_SLOT_BYTES)));
    if (ok) {
        if (threadIdx.x < 6 && blockIdx.x == 0) soa_st(H, stride, threadIdx.x, slot, fp_reduce(tvm_ld(item, (TVM_CLEAR_X + threadIdx.x) * TVM_SLOT_BYTES)));
    } else {
        const uint32_t gbase = threadIdx.x & ~15u;
        g2_jac q0{fp2_from_role(q.x, gbase, 0), fp2_from_role(q.y, gbase, 0), fp2_from_role(q.z, gbase, 0)};
        g2_jac q1{fp2_from_role(q.x, gbase, 1), fp2_from_role(q.y, gbase, 1), fp2_from_role(q.z, gbase, 1)};
        if (threadIdx.x == 0 && blockIdx.x == 0) soa_st_g2(H, stride, slot, clear_cofactor_g2(jac_add(q0, q1)));
    }
#endif
}
// batches that leave wave slots free at 16 lanes per message
__device__ __forceinline__ void team_clear_body(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 lds[4 * TVM_CLEAR_SLOTS * (TVM_SLOT_BYTES / 16)];
    const uint32_t lane16 = threadIdx.x & 15u;
    tvm_lds_char* team = tvm_team_base<TVM_CLEAR_SLOTS>(lds);
    uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool live = i < n;
    if (!live) i = 0;                                             // idle teams recompute message 0 (no stores)
    if (lane16 < 12)                                              // plane j of point 2 i + (0 | 1) -> slots X .. Z | BX .. BZ
        tvm_st(team, (TVM_CLEAR_X + lane16) * TVM_SLOT_BYTES, fp_reduce(soa_ld(M, mstride, lane16 % 6, 2 * (size_t)i + lane16 / 6)));
    (void)tvm_clear_cofactor(team, lane16);
    // Z = 0 marks an exceptional case of the incomplete additions (or a true point at infinity): k_clear_fix, launched behind this kernel, finds
    // such messages by their Z and recomputes them with the complete formulas - in a kernel of its own, so that this one keeps the engine's
    // ~100 registers (the complete formulas need all 512) and several waves fit a SIMD
    if (live && lane16 < 6) soa_st(H, stride, lane16, i, tvm_ld(team, (TVM_CLEAR_X + lane16) * TVM_SLOT_BYTES));
#endif
}
// Small batches (a few hundred messages at most): the engine's program on the ROW executor, a workgroup of four waves per message (rowvm.hpp)
__device__ __forceinline__ void team_clear_rows_body(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 lds[TVM_CLEAR_SLOTS * (TVM_SLOT_BYTES / 16)];
    tvm_lds_char* item = (tvm_lds_char*)(tvm_lds_u32x4*)lds;
    const row_ctx RC = row_ctx_make();
    const uint32_t i = blockIdx.x, t = threadIdx.x;
    if (t < 12) {
        tvm_st(item, (TVM_CLEAR_X + t) * TVM_SLOT_BYTES, fp_reduce(soa_ld(M, mstride, t % 6, 2 * (size_t)i + t / 6)));
    } else if (t < 16) {
        const fp2 cx = fp2_from_const(k::PSI_CX), cy = fp2_from_const(k::PSI_CY);
        tvm_st(item, (TVM_CLEAR_CX + (t - 12)) * TVM_SLOT_BYTES, fp_reduce(fp_select(t < 14, fp_select(t == 12, cx.c0, cx.c1), fp_select(t == 14, cy.c0, cy.c1))));
    } else if (t == 16) {
        tvm_st(item, TVM_CLEAR_zero * TVM_SLOT_BYTES, fp_zero());
    }
    __syncthreads();
    rvm_run<false>(RC, item, TVM_CLEAR_DESC, TVM_CLEAR_SEQ, TVM_CLEAR_NSEQ, tvm_line_sink{});
    // partially reduced, canonical limbs on the way out: k_clear_fix (behind this kernel) finds the exceptional cases by Z = 0
    if (t < 6) soa_st(H, stride, t, i, fp_reduce(tvm_ld(item, (TVM_CLEAR_X + t) * TVM_SLOT_BYTES)));
    (void)n;
#endif
}
// one workgroup per CU (a wave per SIMD) up to 224 items; two per CU - two waves take turns on a SIMD, each ~1.4 x as long - up to 448
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_team_clear_rows(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    team_clear_rows_body(M, mstride, n, H, stride);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_team_clear_rows2(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    team_clear_rows_body(M, mstride, n, H, stride);
}
// Two forms of each engine kernel, same code.  The plain one: ~100 registers, a SIMD takes several waves - for grids beyond one wave per SIMD.
// The SPREAD one declares a whole SIMD's register file (amdgpu_waves_per_eu(1, 1): the register count in the kernel descriptor is raised to
// what keeps a second wave out), for grids of at most one wave per SIMD: the dispatcher fills a CU up to its limits before it opens the
// next, and two latency-bound waves that share a SIMD take twice as long (4 096 messages: 1.19 ms instead of 0.66).  An LDS pad did the
// same until a kernel of the fork stream held LDS on the CU - then the fourth wave no longer fit and started a second round.
__global__ void __launch_bounds__(WAVE) k_team_clear(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    team_clear_body(M, mstride, n, H, stride);
}
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_team_clear_spread(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    team_clear_body(M, mstride, n, H, stride);
}
__global__ void __launch_bounds__(WAVE) k_clear_fix(const uint4* __restrict__ M, size_t mstride, uint32_t n, uint4* __restrict__ H, size_t stride) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fp2 z = soa_ld2(H, stride, 4, i);                       // stored partially reduced: 0 mod p <=> every limb is 0
    if (!(fp_limbs_are_zero(z.c0) & fp_limbs_are_zero(z.c1))) return;
    g2_jac q0 = soa_ld_g2(M, mstride, 2 * (size_t)i), q1 = soa_ld_g2(M, mstride, 2 * (size_t)i + 1);
    soa_st_g2(H, stride, i, clear_cofactor_g2(jac_add(q0, q1)));
}
// the Miller lines of FEW pairs: pairs first .. first + count - 1 -> the step-major line store
__device__ __forceinline__ void team_lines_body(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                uint4* __restrict__ lines) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 lds[4 * TVM_LINES_SLOTS * (TVM_SLOT_BYTES / 16)];
    const uint32_t lane16 = threadIdx.x & 15u;
    tvm_lds_char* team = tvm_team_base<TVM_LINES_SLOTS>(lds);
    uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool live = i < count;
    i = first + (live ? i : 0);
    static_assert(TVM_LINES_PY == TVM_LINES_PX + 1 && TVM_LINES_PZ == TVM_LINES_PX + 2 && TVM_LINES_QX == TVM_LINES_PX + 3, "slot order the prologue relies on");
    if (lane16 < 3) tvm_st(team, (TVM_LINES_PX + lane16) * TVM_SLOT_BYTES, fp_reduce(soa_ld(P, stride, lane16, i)));
    else if (lane16 < 9) tvm_st(team, (TVM_LINES_PX + lane16) * TVM_SLOT_BYTES, fp_reduce(soa_ld(H, stride, lane16 - 3, i)));
    else if (lane16 == 9) tvm_st(team, TVM_LINES_zero * TVM_SLOT_BYTES, fp_zero());
    const bool skip = tvm_slot_is_zero(team, TVM_LINES_PZ) | (tvm_slot_is_zero(team, TVM_LINES_QZ) & tvm_slot_is_zero(team, TVM_LINES_QZ + 1));
    tvm_run<true>(team, lane16, TVM_LINES_DESC, TVM_LINES_SEQ, TVM_LINES_NSEQ, tvm_line_sink{lines, stride, (size_t)i, live, skip});
#endif
}
__global__ void __launch_bounds__(WAVE) k_team_lines(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                     uint4* __restrict__ lines) {
    team_lines_body(P, H, first, count, stride, lines);
}
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_team_lines_spread(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride, uint4* __restrict__ lines) {
    team_lines_body(P, H, first, count, stride, lines);
}
// a workgroup of four waves per pair (rowvm.hpp): the walk of a handful of pairs (fastAggregateVerify: two) in half the time
__device__ __forceinline__ void team_lines_rows_body(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                     uint4* __restrict__ lines) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ bls_u32x4 lds[TVM_LINES_SLOTS * (TVM_SLOT_BYTES / 16)];
    tvm_lds_char* item = (tvm_lds_char*)(tvm_lds_u32x4*)lds;
    const row_ctx RC = row_ctx_make();
    const uint32_t t = threadIdx.x;
    const size_t i = (size_t)first + blockIdx.x;
    if (t < 3) tvm_st(item, (TVM_LINES_PX + t) * TVM_SLOT_BYTES, fp_reduce(soa_ld(P, stride, t, i)));
    else if (t < 9) tvm_st(item, (TVM_LINES_PX + t) * TVM_SLOT_BYTES, fp_reduce(soa_ld(H, stride, t - 3, i)));
    else if (t == 9) tvm_st(item, TVM_LINES_zero * TVM_SLOT_BYTES, fp_zero());
    __syncthreads();
    const bool skip = tvm_slot_is_zero(item, TVM_LINES_PZ) | (tvm_slot_is_zero(item, TVM_LINES_QZ) & tvm_slot_is_zero(item, TVM_LINES_QZ + 1));
    rvm_run<true>(RC, item, TVM_LINES_DESC, TVM_LINES_SEQ, TVM_LINES_NSEQ, tvm_line_sink{lines, stride, i, true, skip});
    (void)count;
#endif
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_team_lines_rows(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride, uint4* __restrict__ lines) {
    team_lines_rows_body(P, H, first, count, stride, lines);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_team_lines_rows2(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride, uint4* __restrict__ lines) {
    team_lines_rows_body(P, H, first, count, stride, lines);
}

// G1 arithmetic has a small live set (a Jacobian point is 42 registers): 256 registers, two waves per SIMD, which fill each
// other's issue gaps (non-multiply VALU instructions issue about twice as fast with a second wave on the SIMD)
// Round 5: the scalar multiplication is ONE hand-allocated assembly statement (tools/gen_pkmul_asm.py -> build/pkmul_asm.inc; curve.hpp's
// jac_mul_u64_w4_body with biased signed digits, jac_dbl_lazy's G1 form and jac_precompute + jac_add_pre, checked by tests/test_asm_loops.py):
// the doubling's multiplier bodies in place on fixed registers, the addition's products through two shared bodies (the loop stays inside the
// instruction cache), the table of 1 .. 8 times the key (with each entry's Z^2, Z^3) in `table`, a buffer of the context with 2 560 contiguous bytes
// per lane (the compiled kernel indexed a table in scratch memory: 16 % of its cycles waiting), a window's entry gathered before the window's four
// doublings.  A lane whose addition met Z3 == 0 (possible only for a
// key outside G1) is flagged and recomputed below with the complete formulas.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_PKMUL_NOASM)
#include "../build/pkmul_asm.inc"
__device__ __forceinline__ uint32_t pkmul_asm(uint64_t r, uint8_t* table, uint32_t off16, uint4* P, uint32_t pstride16, uint32_t lds) {
    uint32_t flag;
    asm volatile(BLS_PKMUL_ASM_BODY : "=v"(flag) : "v"((uint32_t)r), "v"((uint32_t)(r >> 32)), "s"(table), "s"(0u), "v"(off16), "s"(P), "s"(pstride16), "s"(lds) : BLS_PKMUL_ASM_CLOBBERS);
    return flag;
}
#endif
__device__ __forceinline__ void pkmul_body(const uint8_t* __restrict__ sets, uint32_t n, const uint64_t* __restrict__ r, uint4* __restrict__ P,
                                           size_t stride, uint32_t* __restrict__ flags, uint8_t* __restrict__ table) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sets + (size_t)i * 320);
    g1_aff pk = ld_g1a_blst(w);
    const bool inf = aff_is_inf(pk);
    if (inf) atomicOr(flags, 1u);        // BLST_PK_IS_INFINITY -> update() false
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_PKMUL_NOASM)
    if (table) {                                                     // the context's table buffer (MI355_BLS_PKTAB_BYTES per set)
        __shared__ bls_u32x4 key_slot[BLS_LDS_SLOT];
        if (inf) {
            soa_st_g1(P, stride, i, jac_inf<fp>());
        } else {
            fp2_lds_put((bls_lds_u32x4*)key_slot, fp2{pk.x, pk.y});
            if (pkmul_asm(r[i], table, i * 16u, P, (uint32_t)(stride * 16), (uint32_t)(uintptr_t)(bls_lds_u32x4*)key_slot))
                soa_st_g1(P, stride, i, jac_mul_u64_w4_body(pk, r[i]));
        }
        return;
    }
#endif
    g1_jac q = jac_mul_u64_w4_body(pk, r[i]);
    soa_st_g1(P, stride, i, q);
}
__global__ void __launch_bounds__(WAVE, 2) k_pkmul(const uint8_t* __restrict__ sets, uint32_t n, const uint64_t* __restrict__ r, uint4* __restrict__ P,
                                                   size_t stride, uint32_t* __restrict__ flags, uint8_t* __restrict__ table) {
    pkmul_body(sets, n, r, P, stride, flags, table);
}
// The same kernel for grids of at most one wave per SIMD (latency mode, up to 65 536 keys): it declares a whole SIMD's register file, so the
// dispatcher cannot put two of its waves on one SIMD while others are idle (4 096 keys: 64 waves landed on 32 SIMDs and took 1.38 ms instead
// of 0.75 - and held those SIMDs against the cofactor clearing's waves behind it).
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_pkmul_spread(const uint8_t* __restrict__ sets, uint32_t n, const uint64_t* __restrict__ r, uint4* __restrict__ P, size_t stride, uint32_t* __restrict__ flags,
               uint8_t* __restrict__ table) {
    pkmul_body(sets, n, r, P, stride, flags, table);
}

// lines[s] : 6 fp planes (l0.c0,l0.c1,l1.c0,l1.c1,l2.c0,l2.c1), step-major
// Round 5: the 68-step walk is ONE hand-allocated assembly statement (tools/gen_lines_asm.py -> build/lines_asm.inc; formulas, carries and
// reductions of pairing.hpp's miller_dbl_step / miller_add_step, executed and checked lane-for-lane by tests/test_asm_loops.py): T, B and E in
// fixed VGPR blocks, Q and the P-side factors in AGPRs, four multiplier subroutines that read fixed operand slots, the 24 line stores of a
// step issued from the result registers and never waited for.  The compiled prologue (P-side factors, Q in homogeneous form) hands its
// ten Fp values over through five LDS slots.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_LINES_NOASM)
#include "../build/lines_asm.inc"
__device__ __forceinline__ void lines_asm(uint4* lines, uint32_t stride16, uint32_t off16, uint32_t skip, uint32_t lds_a, uint32_t lds_b) {
    asm volatile(BLS_LINES_ASM_BODY : : "s"(lines), "s"(stride16), "v"(off16), "v"(skip), "s"(lds_a), "s"(lds_b) : BLS_LINES_ASM_CLOBBERS);
}
#endif
__global__ void __launch_bounds__(WAVE) k_lines(const uint4* __restrict__ P, const uint4* __restrict__ H, uint32_t first, uint32_t count, size_t stride,
                                                uint4* __restrict__ lines) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    i += first;
    g1_jac p = soa_ld_g1(P, stride, i);
    g2_jac q = soa_ld_g2(H, stride, i);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_LINES_NOASM)
    if (stride * 16 * 24 < ((size_t)1 << 32)) {      // wave-uniform: the loop's row arithmetic is 32-bit (24 rows of a step < 4 GiB: up to 11 M pairs per context)
        __shared__ bls_u32x4 hand_over[4 * BLS_LDS_SLOT];
        const bool skip = jac_is_inf(p) | jac_is_inf(q);
        if (skip) {                                    // a pair with P or Q at infinity contributes 1 (blst skips it): rare, stored here; the loop stores nothing for the lane
            const line_t one = line_one();
            for (int s = 0; s < N_LINES; s++) {
                uint4* b = lines + (size_t)s * 24 * stride;
                soa_st2(b, stride, 0, i, one.l0);
                soa_st2(b, stride, 2, i, one.l1);
                soa_st2(b, stride, 4, i, one.l2);
            }
        }
        const g1_pre pre = g1_precompute(p);
        g2_proj t = g2_to_proj(q);
        bls_lds_u32x4* ho = (bls_lds_u32x4*)hand_over;
        fp2_lds_put(ho, fp2_reduce(t.x));
        fp2_lds_put(ho + BLS_LDS_SLOT, fp2_reduce(t.y));
        fp2_lds_put(ho + 2 * BLS_LDS_SLOT, fp2_reduce(t.z));
        fp2_lds_put(ho + 3 * BLS_LDS_SLOT, fp2{pre.z3, pre.nxz3});
        fp2_lds_put((bls_lds_u32x4*)bls_xchg, fp2{pre.xz, pre.y});       // the multipliers' hand-over slot is free now: every product above is done
        lines_asm(lines, (uint32_t)(stride * 16), i * 16u, skip ? 1u : 0u, (uint32_t)(uintptr_t)ho, (uint32_t)(uintptr_t)(bls_lds_u32x4*)bls_xchg);
        return;                                        // nothing may follow the statement: it leaves m0 / scc / the registers it names clobbered
    }
#endif
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
// Threads of the engine's workgroup: three waves in both context modes.  Phase 1 runs the 144 Fp products of an Fp12 multiplication
// one per thread, phase 2 (c12_phase2_rows) owns one output limb per thread as 12 rows of 16 lanes - it REQUIRES blockDim.x == 192.
constexpr int TAIL_THREADS = 192, TAIL_THREADS_TP = 192;
constexpr int K_TAIL_THREADS = 320;                  // k_tail: the engine's three waves + two: eighteen (+ two idle) rows for the cyclotomic squarings on rows
static_assert(TAIL_THREADS == 192 && TAIL_THREADS_TP == 192, "c12_phase2_rows: 12 coefficient rows of 16 lanes");
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
    if (t >= 12 * 16) return;                                              // k_tail brings two more waves for the cyclotomic squarings (c12_cyc_sqr_rows)
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
// (A 12-row x 16-lane engine with DPP row sums and one reduction per coefficient was built and measured in round 4: faster at best,
// 1.0x .. 1.55x depending on the CU a launch lands on, slower on average - profiles/r04_ab/row_engine.txt.  Removed.)
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

// n squarings in place of register `slot`, which holds a UNITARY value (the hard part of the final exponentiation): Granger-Scott on row arithmetic
// (rowcyc.hpp) - rows 0 .. 17 of the block form the eighteen Fp products along their sixteen lanes, rows 0 .. 11 the new coefficients; two barriers per
// squaring, ~0.7 us where the engine's generic square takes ~2.2.  The block needs 18 rows (288 lanes); the products wait in the engine's product area.
struct cyc_mem_lds {
    c12_lds& S;
    int slot;
    uint32_t* pw;
    __device__ __forceinline__ rw coef(int j, int comp) const {
        const uint32_t l = threadIdx.x & 15u;
        const fp& f = comp ? S.r[slot][j].c1 : S.r[slot][j].c0;
        const uint32_t v = f.l[l < (uint32_t)FP_N ? l : 0u];
        return l < (uint32_t)FP_N ? (rw)v : 0;
    }
    __device__ __forceinline__ rw prod(int r) const { return (rw)pw[r * 16 + (threadIdx.x & 15u)]; }
};
__device__ __noinline__ void c12_cyc_sqr_rows(c12_lds& S, int slot, int n) {
    const row_ctx C = row_ctx_make();
    const int row = (int)(threadIdx.x >> 4), l16 = (int)(threadIdx.x & 15u);
    uint32_t* pw = reinterpret_cast<uint32_t*>(S.w.prod);
    const cyc_mem_lds mem{S, slot, pw};
    const cyc_out_row t = cyc_out_of(row < 12 ? row : 0);
#pragma clang loop unroll(disable)
    for (int i = 0; i < n; i++) {
        if (row < 18) pw[row * 16 + l16] = (uint32_t)cyc_product_row(C, mem, row);
        __syncthreads();
        if (row < 12) {
            const rw o = cyc_output_row(C, mem, row, t);
            fp2& dst = S.r[slot][row >> 1];
            if (l16 < FP_N) ((row & 1) ? dst.c1 : dst.c0).l[l16] = (uint32_t)o;
        }
        __syncthreads();
    }
}
// d = a^x (x < 0, a cyclotomic): square-and-multiply over |x|, then conjugate.  tmp != a.  ROWS (k_tail_rows: a block of 20 rows) takes the runs of
// squarings between the set bits of |x| on row arithmetic.
template <bool ROWS>
__device__ __noinline__ void c12_cyc_exp_x(c12_lds& S, int d, int a, int tmp) {
    c12_copy(S, tmp, a);
    if constexpr (ROWS) {
        int pending = 0;
        for (int bit = 62; bit >= 0; bit--) {
            pending++;
            if ((k::X_ABS >> bit) & 1) {
                c12_cyc_sqr_rows(S, tmp, pending);
                pending = 0;
                c12_mul(S, tmp, tmp, a);
            }
        }
        if (pending) c12_cyc_sqr_rows(S, tmp, pending);
    } else {
        for (int bit = 62; bit >= 0; bit--) {
            c12_sqr(S, tmp, tmp);
            if ((k::X_ABS >> bit) & 1) c12_mul(S, tmp, tmp, a);
        }
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
template <bool ROWS>
__device__ __forceinline__ void tail_body(const uint32_t* __restrict__ L, uint32_t* __restrict__ states, uint32_t kk, int mode,
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
        c12_cyc_exp_x<ROWS>(S, X1, T, X3);            // t^x
        c12_conj(S, X2, T);
        c12_mul(S, A, X1, X2);                  // a = t^(x-1)
        c12_cyc_exp_x<ROWS>(S, X1, A, X3);
        c12_conj(S, X2, A);
        c12_mul(S, A, X1, X2);                  // a = t^((x-1)^2)
        c12_cyc_exp_x<ROWS>(S, X1, A, X3);
        c12_frob(S, X2, A);
        c12_mul(S, B, X1, X2);                  // b = a^(x+p)
        c12_cyc_exp_x<ROWS>(S, X1, B, X3);
        c12_cyc_exp_x<ROWS>(S, X2, X1, X3);           // b^(x^2)
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
        if (lane == 0) printf("   engine phases, ticks of thread 0 (whole kernel %llu): products %llu  barrier %llu  limb sums + reduction %llu  barrier %llu\n",
                              (unsigned long long)(__builtin_amdgcn_s_memtime() - clk0), S.w.prof[0], S.w.prof[1], S.w.prof[2], S.w.prof[3]);
        printf("k_tail mode %d wave %d: %.3f ms simd %u wave_slot %u cu %u se %u raw %x lds_base %u lds_size %u (granules; raw %x)\n", mode, lane >> 6, (double)dr / 1e5, (hwid >> 4) & 3, hwid & 15,
               (hwid >> 8) & 15, (hwid >> 13) & 7, hwid, ldsa & 0xff, (ldsa >> 12) & 0x1ff, ldsa);
    }
#endif
}
// Two forms: k_tail - the engine's three waves, as rounds 3 - 5 (throughput-mode contexts: a block of five waves needs a CU with four free SIMDs at once and
// was measured 0.3 ms per batch SLOWER under three batches in flight); k_tail_rows - two more waves, the cyclotomic squarings of the final exponentiation
// on row arithmetic (latency-mode contexts: every blocking call 0.3 ms shorter, profiles/r06_ab/ab_rowcyc.txt).
__global__ void __launch_bounds__(TAIL_THREADS) k_tail(const uint32_t* __restrict__ L, uint32_t* __restrict__ states, uint32_t kk, int mode,
                                                       uint32_t* __restrict__ gt_out, uint32_t* __restrict__ verdict, uint32_t sstride, int blob) {
    tail_body<false>(L, states, kk, mode, gt_out, verdict, sstride, blob);
}
__global__ void __launch_bounds__(K_TAIL_THREADS) k_tail_rows(const uint32_t* __restrict__ L, uint32_t* __restrict__ states, uint32_t kk, int mode,
                                                       uint32_t* __restrict__ gt_out, uint32_t* __restrict__ verdict, uint32_t sstride, int blob) {
    tail_body<true>(L, states, kk, mode, gt_out, verdict, sstride, blob);
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
// Round 5: the G1 form of the bucket accumulation is ONE hand-allocated assembly statement (tools/gen_msm_asm.py -> build/msm_asm.inc;
// curve.hpp's xyzz_add_aff without its exceptional branches, checked by tests/test_asm_loops.py): the nine multiplier bodies of a mixed
// addition expanded in place on fixed registers (no calls, no operand copies), the NEXT point's record gathered while the current addition
// runs, 230 VGPRs = two waves per SIMD as before.  A lane whose bucket met an exceptional case (addend == +- accumulator, a point at
// infinity: ZZ3 = 0 or the record's flag) comes back flagged and recomputes its bucket with the complete compiled formulas below.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_MSM_NOASM)
#include "../build/msm_asm.inc"
__device__ __forceinline__ uint32_t msm_bucket_asm(const uint32_t* pts, const uint32_t* srt, uint32_t cnt, uint4* buckets, uint32_t ostride16, uint32_t off16) {
    uint32_t flag;
    const uint64_t a = (uint64_t)(uintptr_t)srt;
    asm volatile(BLS_MSM_ASM_BODY : "=v"(flag) : "s"(pts), "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"(cnt), "s"(buckets), "s"(ostride16), "v"(off16) : BLS_MSM_ASM_CLOBBERS);
    return flag;
}
#endif
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
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BLS_MSM_NOASM)
    if constexpr (sizeof(F) == sizeof(fp)) {
        if (!msm_bucket_asm(pts, srt, cnt, buckets, total * 16u, g * 16u)) return;        // stored by the loop; flagged lanes fall through to the complete formulas
    }
#endif
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
// (round 5: the quad team's DPP exchanges instead of team_wave_fp's wave shuffles - every lane holds the same point, so every quad is a team:
// the 112 - 240 dependent doublings of a window sum are what the MSM waits for now)
__device__ __forceinline__ g1_jac g1_dbl_coop(const g1_jac& p) { return jac_dbl_team(p, team_quad_fp<4>{threadIdx.x & 3u}); }
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
// G1, round 6: the window parts -> the windows' values -> ONE Horner walk over the windows [w0, w1), all of it on the limb-parallel row arithmetic
// (rowfp.hpp: a dependent doubling in ~1.4 us where the DPP quad team above takes 4.2).  Wave v of the block sums the nsplit parts of windows v, v + 16, ...;
// wave 0 then walks from window w1 - 1 down: acc = 2^(width) acc + R_w.  acc_in (48 words: X, Y, Z as sixteen limbs each) continues the walk of the window
// group above; the group that ends at window 0 writes the blst image of the result, any other group its accumulator.
__global__ void __launch_bounds__(1024) k_pip_rowtail(const uint32_t* __restrict__ part, uint32_t nsplit, pip_win W, uint32_t w0, uint32_t w1, const uint32_t* __restrict__ acc_in,
                                                      uint32_t* __restrict__ acc_out, uint32_t* __restrict__ out) {
    __shared__ uint32_t sums[64 * 48];
    const row_ctx C = row_ctx_make();
    const uint32_t wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, l16 = threadIdx.x & 15u;
    const bool row0 = (threadIdx.x & 63u) < 16u;
#pragma clang loop unroll(disable)
    for (uint32_t w = w0 + wave; w < w1; w += nwaves) {
        const uint32_t* base = part + (size_t)w * nsplit * (3 * FPW);
        row_g1 acc{row_load(base), row_load(base + FPW), row_load(base + 2 * FPW)};
#pragma clang loop unroll(disable)
        for (uint32_t j = 1; j < nsplit; j++) {
            const uint32_t* q = base + (size_t)j * (3 * FPW);
            acc = row_add(C, acc, row_g1{row_load(q), row_load(q + FPW), row_load(q + 2 * FPW)});
        }
        if (row0) {
            uint32_t* d = sums + (w - w0) * 48;
            d[l16] = (uint32_t)acc.x; d[16 + l16] = (uint32_t)acc.y; d[32 + l16] = (uint32_t)acc.z;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    row_g1 acc;
    uint32_t w = w1;
    if (acc_in) {
        acc = row_g1{(rw)acc_in[l16], (rw)acc_in[16 + l16], (rw)acc_in[32 + l16]};
    } else {
        w--;
        const uint32_t* d = sums + (w - w0) * 48;
        acc = row_g1{(rw)d[l16], (rw)d[16 + l16], (rw)d[32 + l16]};
    }
#pragma clang loop unroll(disable)
    while (w > w0) {
        w--;
        const uint32_t sh = pip_off(W, w + 1) - pip_off(W, w);
#pragma clang loop unroll(disable)
        for (uint32_t i = 0; i < sh; i++) acc = row_dbl(C, acc);
        const uint32_t* d = sums + (w - w0) * 48;
        acc = row_add(C, acc, row_g1{(rw)d[l16], (rw)d[16 + l16], (rw)d[32 + l16]});
    }
    if (w0 == 0) {
        const g1_jac r{fp_reduce(row_to_fp(acc.x)), fp_reduce(row_to_fp(acc.y)), fp_reduce(row_to_fp(acc.z))};
        if (threadIdx.x == 0) st_jac_blst(out, r);
    } else if (row0) {
        acc_out[l16] = (uint32_t)acc.x; acc_out[16 + l16] = (uint32_t)acc.y; acc_out[32 + l16] = (uint32_t)acc.z;
    }
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

// small glue kernels of the host layer (host_api.inc)
__global__ void k_or_flag(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && src[0]) atomicOr(dst, src[0]);
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

}  // namespace

// The host side - contexts, the C ABI of include/blscurve_mi355x.h, the slice pipeline, the multi-device drivers - is the second file of this
// translation unit (one TU: the kernels above are in an anonymous namespace and are launched with <<<>>> from there).
#include "host_api.inc"
