// The lane-team engine of the latency path (round 6): hash-to-G2's cofactor clearing and the 68-step Miller walk of ONE message / pair on
// a team of 16 lanes (one DPP row), for calls that cannot fill the chip with a lane per item - fastAggregateVerify, one signature, batches
// of up to a few thousand sets (reference: bls_sig_min_pubkey.nim:234-258, blst_min_pubkey_sig_core.nim:269-297, blst_abi.nim:383, :455).
//
// The team's values live in LDS as 64-byte Fp slots (partially reduced: |v| < 0.51 p, canonical limbs) and the point formulas are DATA
// (tools/teamvm.py -> build/teamvm_tables.inc): a program is a sequence of rounds, a round gives every lane one descriptor
//     v = S[a] * S[b];   out = reduce(c0 v + c1 v^1 + c2 v^2 + c3 v^3 + ct S[t]);   S[dst] = out
// with v^k the product of lane (l xor k) of the same quad (three DPP quad permutes).  An Fp2 product is a quad (re = v0 - v1, im = v2 + v3),
// a lazily reduced a b - c d costs nothing extra, every small multiple of the formulas is an integer coefficient folded into the
// reduction's own 64-bit chain.  One instruction stream for every round of every formula: no selects, no shuffles, and the kernel
// holds three Fp values and a descriptor, so nothing spills (the compiled team formulas of rounds 1-5 kept every intermediate in the
// registers of every lane: ~700 spilled registers, ~5 300 cycles per round for a 1 840-cycle product).
// tvm_post is host-callable: tests/host_emu runs whole programs on sixteen emulated lanes under the bounds tracker.
#pragma once
#include "fp.hpp"

namespace bls {

constexpr uint32_t TVM_F_LINEAR = 1u << 16, TVM_F_GSTORE = 1u << 17, TVM_STEP_SHIFT = 20, TVM_NO_PLANE = 15;
constexpr int TVM_TEAM = 16, TVM_SLOT_BYTES = 80;      // 64 bytes of limbs + 16 of padding (LDS banks: tools/teamvm.py SLOT_BYTES)

BLS_HD int32_t tvm_sbyte(uint32_t w, int k) { return (int32_t)(w << (24 - 8 * k)) >> 24; }

// reduce(c0 v0 + c1 v1 + c2 v2 + c3 v3 + ct t): fp_reduce with the linear combination folded into its chain.  Operands: canonical limbs
// (limbs 0..12 in [0, 2^28)), |v| < 2 p (products) or 0.51 p (slots); sum of |c| <= 64 (asserted when the tables are built).  The quotient
// is estimated from the top limbs alone: the lower limbs of canonical values add less than sum|c| 2^364 < 2^-11 p, so |out| < 0.51 p.
BLS_HD fp tvm_post(const fp& v0, const fp& v1, const fp& v2, const fp& v3, const fp& t, int32_t c0, int32_t c1, int32_t c2, int32_t c3, int32_t ct) {
    BLS_REQUIRE(BLS_LB(v0) == 0 && BLS_LB(v1) == 0 && BLS_LB(v2) == 0 && BLS_LB(v3) == 0 && BLS_LB(t) == 0, "tvm_post: canonical operands");
    BLS_REQUIRE((uint64_t)(c0 < 0 ? -c0 : c0) * BLS_VB(v0) + (uint64_t)(c1 < 0 ? -c1 : c1) * BLS_VB(v1) + (uint64_t)(c2 < 0 ? -c2 : c2) * BLS_VB(v2) +
                        (uint64_t)(c3 < 0 ? -c3 : c3) * BLS_VB(v3) + (uint64_t)(ct < 0 ? -ct : ct) * BLS_VB(t) <= 1024, "tvm_post value bound");
    const int64_t RECIP = 10322735;                       // round(2^40 / (p / 2^364)), as fp_reduce
    const int32_t top = c0 * (int32_t)v0.l[FP_N - 1] + c1 * (int32_t)v1.l[FP_N - 1] + c2 * (int32_t)v2.l[FP_N - 1] + c3 * (int32_t)v3.l[FP_N - 1] +
                        ct * (int32_t)t.l[FP_N - 1];
    const int32_t nq = -(int32_t)(((int64_t)top * RECIP + (1ll << 39)) >> 40);
    fp r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < FP_N; i++) {
        acc = bls_mac(acc, (int32_t)v0.l[i], c0);
        acc = bls_mac(acc, (int32_t)v1.l[i], c1);
        acc = bls_mac(acc, (int32_t)v2.l[i], c2);
        acc = bls_mac(acc, (int32_t)v3.l[i], c3);
        acc = bls_mac(acc, (int32_t)t.l[i], ct);
        acc = bls_mac_c(acc, nq, k::P[i]);
        if (i < FP_N - 1) {
            r.l[i] = (uint32_t)acc & FP_MASK;
            acc >>= 28;
        } else {
            r.l[i] = (uint32_t)acc;
        }
    }
    BLS_SET_VB(r, 1);
    BLS_SET_LB(r, 0);
    return r;
}

// one lane's product of a round (the operands are slots: canonical, |v| < 0.51 p)
BLS_HD fp tvm_product(const fp& a, const fp& b) {
    BLS_REQUIRE(BLS_LB(a) == 0 && BLS_LB(b) == 0 && BLS_VB(a) <= 2 && BLS_VB(b) <= 2, "tvm_product operands");
    BLS_COUNT_MADS(392);
    return fp_mul_core(a, b);
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) bls_u32x4 tvm_lds_u32x4;
typedef __attribute__((address_space(3))) char tvm_lds_char;

__device__ __forceinline__ fp tvm_ld(const tvm_lds_char* team, uint32_t off) {
    const tvm_lds_u32x4* q = reinterpret_cast<const tvm_lds_u32x4*>(team + off);
    bls_u32x4 x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
    return fp{{x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y}};
}
__device__ __forceinline__ void tvm_st(tvm_lds_char* team, uint32_t off, const fp& a) {
    tvm_lds_u32x4* q = reinterpret_cast<tvm_lds_u32x4*>(team + off);
    q[0] = bls_u32x4{a.l[0], a.l[1], a.l[2], a.l[3]};
    q[1] = bls_u32x4{a.l[4], a.l[5], a.l[6], a.l[7]};
    q[2] = bls_u32x4{a.l[8], a.l[9], a.l[10], a.l[11]};
    q[3] = bls_u32x4{a.l[12], a.l[13], 0u, 0u};
}
template <int CTRL>
__device__ __forceinline__ fp tvm_quad(const fp& a) {
    fp r;
#pragma unroll
    for (int i = 0; i < FP_N; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], CTRL, 0xf, 0xf, true);
    return r;
}

// Where a round's line planes go (Miller walk only): step-major line store, plane p of pair i at lines[((step * 6 + p) * 4 + q) * stride + i].
struct tvm_line_sink {
    uint4* lines;
    size_t stride, pair;
    bool live, skip;          // live: this team has a pair; skip: P or Q at infinity - the pair's lines are 1 (blst skips such pairs)
};

// Runs `nseq` rounds on this lane's team.  team: the team's LDS region; lane16: the lane's place in the team; desc / seq: the program.
template <bool LINES>
__device__ __forceinline__ void tvm_run(tvm_lds_char* team, uint32_t lane16, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ seq, uint32_t nseq,
                                        const tvm_line_sink& sink) {
    const uint4* dtab = reinterpret_cast<const uint4*>(desc) + lane16;
    uint32_t e = seq[0], en = seq[1];
    uint4 d = dtab[(e & 0xffffu) * TVM_TEAM];
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < nseq; i++) {
        // two rounds ahead: the sequence word (a scalar load; the tables carry two entries of padding), one round ahead: the descriptor -
        // both are on their way during the product and neither is waited for before the next round needs it
        const uint32_t enn = seq[i + 2];
        const uint4 dn = dtab[(en & 0xffffu) * TVM_TEAM];
        // all three operands are requested at once (one LDS latency per round, not three): b is the zero slot in a linear round
        const fp a = tvm_ld(team, d.x & 0xffffu);
        const fp b = tvm_ld(team, d.x >> 16);
        const fp t = tvm_ld(team, d.y & 0xffffu);
        fp v;
        if (e & TVM_F_LINEAR) {
            v = a;
        } else {
            v = fp_mul_core(a, b);
        }
        const fp v1 = tvm_quad<0xb1>(v), v2 = tvm_quad<0x4e>(v), v3 = tvm_quad<0x1b>(v);      // lanes l ^ 1, l ^ 2, l ^ 3 of the quad
        fp o = tvm_post(v, v1, v2, v3, t, tvm_sbyte(d.z, 0), tvm_sbyte(d.z, 1), tvm_sbyte(d.z, 2), tvm_sbyte(d.z, 3), tvm_sbyte(d.w, 0));
        tvm_st(team, d.y >> 16, o);
        if (LINES && (e & TVM_F_GSTORE)) {
            const uint32_t plane = (d.w >> 8) & 0xfu;
            if (plane != TVM_NO_PLANE && sink.live) {
                if (sink.skip) o = plane == 0 ? fp_one() : fp_zero();
                uint4* b = sink.lines + (size_t)((e >> TVM_STEP_SHIFT) & 0xffu) * 24 * sink.stride + (size_t)plane * 4 * sink.stride + sink.pair;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    b[(size_t)q * sink.stride] = q < 3 ? make_uint4(o.l[4 * q], o.l[4 * q + 1], o.l[4 * q + 2], o.l[4 * q + 3]) : make_uint4(o.l[12], o.l[13], 0u, 0u);
            }
        }
        e = en;
        en = enn;
        d = dn;
    }
}
#endif

// Host form (tests/host_emu): the same rounds on sixteen emulated lanes; S: the team's slots.  on_line(step, plane, value) per line plane.
template <class OnLine>
inline void tvm_run_host(fp* S, const uint32_t* desc, const uint32_t* seq, uint32_t nseq, OnLine&& on_line) {
    for (uint32_t i = 0; i < nseq; i++) {
        const uint32_t e = seq[i];
        const uint32_t* d = desc + (size_t)(e & 0xffffu) * TVM_TEAM * 4;
        fp v[TVM_TEAM], o[TVM_TEAM];
        for (int l = 0; l < TVM_TEAM; l++) {
            const fp& a = S[(d[4 * l] & 0xffffu) / TVM_SLOT_BYTES];
            v[l] = (e & TVM_F_LINEAR) ? a : tvm_product(a, S[(d[4 * l] >> 16) / TVM_SLOT_BYTES]);
        }
        for (int l = 0; l < TVM_TEAM; l++) {
            const uint32_t* w = d + 4 * l;
            o[l] = tvm_post(v[l], v[l ^ 1], v[l ^ 2], v[l ^ 3], S[(w[1] & 0xffffu) / TVM_SLOT_BYTES], tvm_sbyte(w[2], 0), tvm_sbyte(w[2], 1), tvm_sbyte(w[2], 2),
                            tvm_sbyte(w[2], 3), tvm_sbyte(w[3], 0));
        }
        for (int l = 0; l < TVM_TEAM; l++) {
            const uint32_t* w = d + 4 * l;
            S[(w[1] >> 16) / TVM_SLOT_BYTES] = o[l];
            const uint32_t plane = (w[3] >> 8) & 0xfu;
            if ((e & TVM_F_GSTORE) && plane != TVM_NO_PLANE) on_line((e >> TVM_STEP_SHIFT) & 0xffu, plane, o[l]);
        }
    }
}

}  // namespace bls
