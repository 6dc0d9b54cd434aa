// The lane-team engine's programs (teamvm.hpp, tables from tools/teamvm.py) executed on ROWS: the sixteen "lanes" of a team become the sixteen rows of four waves
// (a workgroup of 256 lanes per item), every slot value lies along a row (limb l in lane l: rowfp.hpp), and a round's product is row_mul - ~0.44 us where the
// lane of the team engine takes ~0.9.  Same slots (the 14 + 2 zero words of a slot ARE a row), same descriptors, same sequences:
//     v = S[a] * S[b];   out = reduce(c0 v + c1 v^1 + c2 v^2 + c3 v^3 + ct S[t]);   S[dst] = out
// with v^k the product of row (r xor k) of the same wave (the team's quad = the four rows of a wave).  For calls with at most a couple of hundred items
// (fastAggregateVerify, one signature, small batches): four waves per item instead of a quarter of one.
// The round body is written on `rw` (rowfp.hpp): the device runs it per lane, tests/host_emu on 64 emulated lanes per wave with every bound asserted.
#pragma once
#include "rowcyc.hpp"
#include "teamvm.hpp"

namespace bls {

struct rvm_desc { rw x, y, z, w; };          // the row's descriptor of the round, in every lane of the row

// the linear combination of a round and its partial reduction (tvm_post's arithmetic, limb-parallel): |out| < 0.51 p + a little, limbs 0..12 within
// [-2^8, 2^28 + 2^8] (`tight`: one more carry pass, [-1, 2^28 + 1] - values that leave the engine), limb 13 signed, lanes 14, 15 zero
ROW_FN rw rvm_post(const row_ctx& C, const rw& v, const rw& t, const rvm_desc& d, bool tight) {
    const rw v1 = row_xrow<1>(v), v2 = row_xrow<2>(v), v3 = row_xrow<3>(v);
    rw64 acc = row_zero64();
    acc = row_mad(v, row_sbyte(d.z, 0), acc);
    acc = row_mad(v1, row_sbyte(d.z, 1), acc);
    acc = row_mad(v2, row_sbyte(d.z, 2), acc);
    acc = row_mad(v3, row_sbyte(d.z, 3), acc);
    acc = row_mad(t, row_sbyte(d.w, 0), acc);
    // quotient from the top limb (fp_reduce's estimate: the limbs below add less than 2^-10 p to a value of at most 66 p), one linear carry pass from the
    // 64-bit sums (|.| < 2^36): the top limb keeps what is above it
    rw x = row_reduce64(C, acc);
    if (tight) x = row_norm(C, x);
    return x;
}

// One round for the rows of this wave.  mem.load(byte offsets) -> the slot's limb of this lane.  Returns the value to store at row_hi16(d.y).
template <class Mem>
ROW_FN rw rvm_round(const row_ctx& C, const Mem& mem, const rvm_desc& d, bool linear, bool tight) {
    const rw a = mem.load(row_lo16(d.x)), t = mem.load(row_lo16(d.y));
    rw v = a;
    if (!linear) v = row_mul(C, a, mem.load(row_hi16(d.x)));
    return rvm_post(C, v, t, d, tight);
}

#if defined(BLS_ROW_EMU)
// tests/host_emu: the slots of ONE item as fp values (the team engine's own layout); four emulated waves per round
struct rvm_mem_host {
    const fp* S;
    int wave;
    rw load(const rw& off) const {
        rw r;
        for (int i = 0; i < 64; i++) r.l[i] = (i & 15) < FP_N ? (int32_t)S[off.l[i] / TVM_SLOT_BYTES].l[i & 15] : 0;
        return r;
    }
};
template <class OnLine>
inline void rvm_run_host(fp* S, const uint32_t* desc, const uint32_t* seq, uint32_t nseq, OnLine&& on_line) {
    const row_ctx C = row_ctx_make();
    for (uint32_t i = 0; i < nseq; i++) {
        const uint32_t e = seq[i];
        const uint32_t* dd = desc + (size_t)(e & 0xffffu) * TVM_TEAM * 4;
        rw out[4];
        for (int wv = 0; wv < 4; wv++) {
            rvm_desc d;
            for (int l = 0; l < 64; l++) {
                const uint32_t* w = dd + 4 * (4 * wv + (l >> 4));
                d.x.l[l] = (int32_t)w[0]; d.y.l[l] = (int32_t)w[1]; d.z.l[l] = (int32_t)w[2]; d.w.l[l] = (int32_t)w[3];
            }
            out[wv] = rvm_round(C, rvm_mem_host{S, wv}, d, (e & TVM_F_LINEAR) != 0, (e & TVM_F_GSTORE) != 0);
        }
        for (int r = 0; r < TVM_TEAM; r++) {
            const uint32_t* w = dd + 4 * r;
            fp o;
            for (int k = 0; k < FP_N; k++) o.l[k] = (uint32_t)out[r >> 2].l[(r & 3) * 16 + k];
            BLS_SET_VB(o, 1);
            BLS_SET_LB(o, 1);
            S[(w[1] >> 16) / TVM_SLOT_BYTES] = o;
            const uint32_t plane = (w[3] >> 8) & 0xfu;
            if ((e & TVM_F_GSTORE) && plane != TVM_NO_PLANE) on_line((e >> TVM_STEP_SHIFT) & 0xffu, plane, o);
        }
    }
}
#elif defined(__HIP_DEVICE_COMPILE__)
struct rvm_mem_lds {
    const tvm_lds_char* item;          // the item's slots, this lane's limb: item + 4 * (lane & 15) already added
    __device__ __forceinline__ rw load(rw off) const { return (rw) * reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(item + off); }
};
// Runs `nseq` rounds for one item on the block's sixteen rows (256 lanes).  item: the item's slot region in LDS.
template <bool LINES>
__device__ __forceinline__ void rvm_run(const row_ctx& C, tvm_lds_char* item, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ seq, uint32_t nseq,
                                        const tvm_line_sink& sink) {
    const uint32_t l16 = threadIdx.x & 15u, row = (threadIdx.x >> 4) & 15u;
    const uint4* dtab = reinterpret_cast<const uint4*>(desc) + row;
    tvm_lds_char* mine = item + 4 * l16;
    const rvm_mem_lds mem{mine};
    uint32_t e = seq[0], en = seq[1];
    uint4 d = dtab[(e & 0xffffu) * TVM_TEAM];
#pragma clang loop unroll(disable)
    for (uint32_t i = 0; i < nseq; i++) {
        const uint32_t enn = seq[i + 2];
        const uint4 dn = dtab[(en & 0xffffu) * TVM_TEAM];
        const bool gstore = LINES && (e & TVM_F_GSTORE);
        rw o = rvm_round(C, mem, rvm_desc{(rw)d.x, (rw)d.y, (rw)d.z, (rw)d.w}, (e & TVM_F_LINEAR) != 0, gstore);
        __syncthreads();                                  // every row has read this round's operands
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(mine + (d.y >> 16)) = (uint32_t)o;
        if (gstore) {
            const uint32_t plane = (d.w >> 8) & 0xfu;
            if (plane != TVM_NO_PLANE && sink.live) {
                if (sink.skip) o = plane == 0 ? row_from_fp(fp_one()) : 0;
                uint32_t* b = reinterpret_cast<uint32_t*>(sink.lines + (size_t)((e >> TVM_STEP_SHIFT) & 0xffu) * 24 * sink.stride + (size_t)plane * 4 * sink.stride + sink.pair +
                                                          (size_t)(l16 >> 2) * sink.stride);
                b[l16 & 3u] = (uint32_t)o;
            }
        }
        __syncthreads();                                  // ... and sees this round's results
        e = en;
        en = enn;
        d = dn;
    }
}
#endif

}  // namespace bls
