// Micro-benchmark: issue cost of the integer instructions a 384-bit Montgomery multiplier is made of.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 7, d = b + 11;
    uint64_t x0 = a, x1 = b, x2 = c, x3 = d;
    uint32_t y0 = a, y1 = b, y2 = c, y3 = d;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // 4 independent chains of v_mad_u64_u32
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 1) {  // dependent chain
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0"
                               : "+v"(x0) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 2) {  // v_add_u32 independent
            REP16(asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a));)
        } else if (KIND == 3) {  // v_lshl_add_u64
            REP16(asm volatile("v_lshl_add_u64 %0, %4, 0, %0\n v_lshl_add_u64 %1, %4, 0, %1\n v_lshl_add_u64 %2, %4, 0, %2\n v_lshl_add_u64 %3, %4, 0, %3"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x0));)
        } else if (KIND == 4) {  // v_mul_lo_u32
            REP16(asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a));)
        } else if (KIND == 5) {  // v_mul_hi_u32
            REP16(asm volatile("v_mul_hi_u32 %0, %4, %0\n v_mul_hi_u32 %1, %4, %1\n v_mul_hi_u32 %2, %4, %2\n v_mul_hi_u32 %3, %4, %3"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a));)
        } else if (KIND == 6) {  // add with carry chain: v_add_co_u32 + v_addc_co_u32 x3
            REP16(asm volatile("v_add_co_u32 %0, vcc, %4, %0\n v_addc_co_u32 %1, vcc, %4, %1, vcc\n v_addc_co_u32 %2, vcc, %4, %2, vcc\n v_addc_co_u32 %3, vcc, %4, %3, vcc"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a) : "vcc");)
        } else if (KIND == 7) {  // v_mad_u32_u24 (full-rate 24-bit mad)
            REP16(asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b));)
        } else if (KIND == 8) {  // mad + addc pair (carry-out consumed)
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc"
                               : "+v"(x0), "+v"(x1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 9) {  // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                               : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a));)
        } else if (KIND == 10) {  // double FMA
            double f0 = (double)y0, f1 = (double)y1, f2 = (double)y2, f3 = (double)y3, fa = 1.0000001;
            REP16(asm volatile("v_fma_f64 %0, %4, %0, %0\n v_fma_f64 %1, %4, %1, %1\n v_fma_f64 %2, %4, %2, %2\n v_fma_f64 %3, %4, %3, %3"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa));)
            y0 += (uint32_t)f0 + (uint32_t)f1 + (uint32_t)f2 + (uint32_t)f3;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3) + y0 + y1 + y2 + y3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int blocks, int threads) {
    uint32_t* out; uint64_t* cyc; int iters = 2000;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    k<KIND><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    double ninstr = (double)iters * 64;
    printf("%-28s blocks=%5d thr=%4d  memtime-ticks/instr(wave0)=%6.2f  wall ns/instr/wave=%7.3f\n", name, blocks, threads, (double)c0 / ninstr, ms * 1e6 / ninstr);
    hipFree(out); hipFree(cyc);
}
#define ALL(K, name) run<K>(name, 1024, 64); run<K>(name, 1024, 256); run<K>(name, 2048, 256);
int main() {
    ALL(0, "mad_u64_u32 indep") ALL(1, "mad_u64_u32 dep") ALL(2, "add_u32") ALL(3, "lshl_add_u64") ALL(4, "mul_lo_u32") ALL(5, "mul_hi_u32")
    ALL(6, "add_co/addc chain") ALL(7, "mad_u32_u24") ALL(8, "mad_u64+addc") ALL(9, "mov_b32") ALL(10, "fma_f64")
    return 0;
}
