// Micro-benchmark, the CEILING of the integer multiply-add roofline for this arithmetic: lazily reduced dot products a0 b0 + a1 b1
// (fp_dot2_core: 588 v_mad_i64_i32 + 68 bookkeeping instructions, half an Fp2 product) back to back with zero caller code - no
// argument moves, no additions, no carries, no memory - in a loop small enough for the instruction cache, one wave per SIMD on every
// SIMD.  What it reaches, as a fraction of the nominal peak (one multiply-add per SIMD lane per 4 cycles at 2.4 GHz), is the most
// ANY kernel built on this multiplier can reach: bench.py prints it as roofline.int_mad.ceiling.
// Build: python3 nim-blscurve_amd/tools/gen_lineprod_asm.py --ubench-dot2 4 -o tools/ubench_fp2chain.inc
//        hipcc -O3 --offload-arch=gfx950 tools/ubench_fp2chain.hip -o tools/ubench_fp2chain.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "ubench_fp2chain.inc"
constexpr int BODIES = 4, MADS_PER_BODY = 588, INSTR_PER_BODY = 657;
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* stamps, int iters) {
    // operands: any 28-bit limbs do (the multiply-adds do not care what the values mean)
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    asm volatile(".p2align 6\n"
                 "v_mov_b32_e32 v0, %0\n v_and_b32_e32 v0, 0xfffffff, v0\n"
                 "v_mov_b32_e32 v1, v0\n v_mov_b32_e32 v2, v0\n v_mov_b32_e32 v3, v0\n v_mov_b32_e32 v4, v0\n v_mov_b32_e32 v5, v0\n v_mov_b32_e32 v6, v0\n v_mov_b32_e32 v7, v0\n"
                 "v_mov_b32_e32 v8, v0\n v_mov_b32_e32 v9, v0\n v_mov_b32_e32 v10, v0\n v_mov_b32_e32 v11, v0\n v_mov_b32_e32 v12, v0\n v_mov_b32_e32 v13, v0\n"
                 "v_mov_b32_e32 v14, v0\n v_mov_b32_e32 v15, v0\n v_mov_b32_e32 v16, v0\n v_mov_b32_e32 v17, v0\n v_mov_b32_e32 v18, v0\n v_mov_b32_e32 v19, v0\n v_mov_b32_e32 v20, v0\n"
                 "v_mov_b32_e32 v21, v0\n v_mov_b32_e32 v22, v0\n v_mov_b32_e32 v23, v0\n v_mov_b32_e32 v24, v0\n v_mov_b32_e32 v25, v0\n v_mov_b32_e32 v26, v0\n v_mov_b32_e32 v27, v0\n"
                 "v_mov_b32_e32 v28, v0\n v_mov_b32_e32 v29, v0\n v_mov_b32_e32 v30, v0\n v_mov_b32_e32 v31, v0\n v_mov_b32_e32 v32, v0\n v_mov_b32_e32 v33, v0\n v_mov_b32_e32 v34, v0\n"
                 "v_mov_b32_e32 v35, v0\n v_mov_b32_e32 v36, v0\n v_mov_b32_e32 v37, v0\n v_mov_b32_e32 v38, v0\n v_mov_b32_e32 v39, v0\n v_mov_b32_e32 v40, v0\n v_mov_b32_e32 v41, v0\n"
                 "v_mov_b32_e32 v42, v0\n v_mov_b32_e32 v43, v0\n v_mov_b32_e32 v44, v0\n v_mov_b32_e32 v45, v0\n v_mov_b32_e32 v46, v0\n v_mov_b32_e32 v47, v0\n v_mov_b32_e32 v48, v0\n"
                 "v_mov_b32_e32 v49, v0\n v_mov_b32_e32 v50, v0\n v_mov_b32_e32 v51, v0\n v_mov_b32_e32 v52, v0\n v_mov_b32_e32 v53, v0\n v_mov_b32_e32 v54, v0\n v_mov_b32_e32 v55, v0\n"
                 : : "v"(seed) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
                 "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
                 "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile(".p2align 6\n" UBENCH_BODY : : : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22",
                     "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                     "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
                     "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t res;
    asm volatile("v_mov_b32_e32 %0, v0" : "=v"(res));
    out[blockIdx.x * 64 + threadIdx.x] = res;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
    // usage: ubench_fp2chain.bin [iters [only [waves_per_simd]]]   (bench.py runs "4000 0": ~20 ms on every SIMD of the chip, one line of output)
    const int iters_arg = argc > 1 ? atoi(argv[1]) : 20000;
    int nsimd = 1024;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) == hipSuccess) nsimd = 4 * prop.multiProcessorCount;
    }
    const int only = argc > 2 ? atoi(argv[2]) : -1;          // 0: the whole chip only
    const int per_simd = argc > 3 ? atoi(argv[3]) : 1;       // waves per SIMD (the kernel needs 72 registers: up to 7 fit)
    for (int blocks : {nsimd * per_simd, nsimd / 8}) {
        if (only == 0 && blocks != nsimd * per_simd) continue;
        uint32_t* out; uint64_t* st; int iters = iters_arg;
        (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&st, blocks * 16);
        k<<<blocks, 64>>>(out, st, 200);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k<<<blocks, 64>>>(out, st, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[2]; (void)hipMemcpy(h, st + 2 * (blocks / 2), 16, hipMemcpyDeviceToHost);
        double instr = (double)iters * BODIES * INSTR_PER_BODY, mads = (double)iters * BODIES * MADS_PER_BODY;
        double rate = mads * 64.0 * blocks / (ms * 1e-3);                 // multiply-adds per second, whole launch
        double peak = (double)nsimd * 64 * 2.4e9 / 4;
        if (blocks > nsimd) {            // several waves per SIMD: cycles per instruction PER SIMD
            printf("dot2 chain, %4d waves (%d per SIMD): cycles/instr per wave %.3f = %.3f per SIMD  in-kernel clock %.3f GHz  multiply-adds/s %.2f T  (%.3f of the nominal peak %.1f T)\n", blocks, per_simd,
                   (double)h[0] / instr, (double)h[0] / instr / per_simd, (double)h[0] / ((double)h[1] * 10.0), rate / 1e12, rate / peak, peak / 1e12);
            continue;
        }
        printf("dot2 chain, %4d waves: cycles/instr %.3f  in-kernel clock %.3f GHz  multiply-adds/s %.2f T  (%.3f of the nominal peak %.1f T when all 1024 SIMDs run)\n", blocks,
               (double)h[0] / instr, (double)h[0] / ((double)h[1] * 10.0), rate / 1e12, rate * ((double)nsimd / blocks) / peak, peak / 1e12);
    }
    return 0;
}
