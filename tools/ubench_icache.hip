// Micro-benchmark: does a LOOP BODY larger than the instruction cache cost issue slots at one wave per SIMD?  Bodies of 1 K .. 32 K
// dependent v_mad_i64_i32 (8 KB .. 256 KB of code; the cache is 64 KB shared by two CUs), every SIMD of the chip busy (1024 waves)
// or an eighth of it (128), cycles per instruction from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_icache.hip -o tools/ubench_icache.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1K(x) R4(R256(x))
#define MAD "v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
template <int KB>
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a = (threadIdx.x * 2654435761u + 1) & 0x0fffffffu, b = (a ^ 0x9e3779b9u) & 0x0fffffffu;
    int64_t x0 = a;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KB >= 1) asm volatile(".p2align 6\n" R1K(MAD) : "+v"(x0) : "v"(a), "v"(b) : "vcc");
        if (KB >= 2) asm volatile(R1K(MAD) : "+v"(x0) : "v"(a), "v"(b) : "vcc");
        if (KB >= 4) { asm volatile(R1K(MAD) R1K(MAD) : "+v"(x0) : "v"(a), "v"(b) : "vcc"); }
        if (KB >= 8) { asm volatile(R4(R1K(MAD)) : "+v"(x0) : "v"(a), "v"(b) : "vcc"); }
        if (KB >= 16) { asm volatile(R4(R1K(MAD)) R4(R1K(MAD)) : "+v"(x0) : "v"(a), "v"(b) : "vcc"); }
        if (KB >= 32) { asm volatile(R16(R1K(MAD)) : "+v"(x0) : "v"(a), "v"(b) : "vcc"); }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KB>
void run(int blocks) {
    uint32_t* out; uint64_t* cyc; int iters = 65536 / KB / 4;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    k<KB><<<blocks, 64>>>(out, cyc, 2);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<KB><<<blocks, 64>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t c0; (void)hipMemcpy(&c0, cyc + blocks / 2, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * KB * 1024;
    printf("loop body %3d K multiply-adds = %4d KB of code, %4d waves: cycles/instr=%6.3f  wall ns/instr=%6.3f\n", KB, KB * 8, blocks, (double)c0 / n, ms * 1e6 / n);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int blocks : {128, 1024}) { run<1>(blocks); run<2>(blocks); run<4>(blocks); run<8>(blocks); run<16>(blocks); run<32>(blocks); }
    return 0;
}
