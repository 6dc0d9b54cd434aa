// Micro-benchmark: does the 8-byte alignment of a stream of 8-byte VALU instructions matter on gfx950?
// One asm block: .p2align 6, then OFF/4 four-byte s_nop, then 256 v_mad_u64_u32 (8 bytes each), looped.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_align.hip -o tools/ubench_align.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
template <int NOPS>
__global__ void k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u;
    uint64_t x0 = a, x1 = b;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (NOPS == 0)
            asm volatile(".p2align 6\n" R256("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else if (NOPS == 1)
            asm volatile(".p2align 6\n s_nop 0\n" R256("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else if (NOPS == 2)
            asm volatile(".p2align 6\n s_nop 0\n s_nop 0\n" R256("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else if (NOPS == 4)   // the 4-byte v_mov re-encoded as VOP3 (8 bytes): parity never flips
            asm volatile(".p2align 6\n" R16(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") "v_mov_b32_e64 %2, %2\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else if (NOPS == 5)   // a 4-byte scalar instruction re-aligned with a 4-byte s_nop
            asm volatile(".p2align 6\n" R16(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") "s_waitcnt lgkmcnt(0)\n s_nop 0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else if (NOPS == 6)   // a 4-byte scalar instruction alone: parity flips
            asm volatile(".p2align 6\n" R16(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") "s_waitcnt lgkmcnt(0)\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        else  // every 16th instruction followed by one 4-byte v_mov: parity flips back and forth
            asm volatile(".p2align 6\n" R16(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n") "v_mov_b32 %2, %2\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x0 + (uint32_t)x1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NOPS>
void run(const char* name) {
    uint32_t* out; uint64_t* cyc; int iters = 2000;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
    k<NOPS><<<1024, 64>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NOPS><<<1024, 64>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 256;
    printf("%-44s ticks/mad=%6.3f  wall ns/mad=%7.3f\n", name, (double)c0 / n, ms * 1e6 / n);
}
int main() {
    for (int r = 0; r < 2; r++) {
        run<0>("256 mads, stream 8-byte aligned");
        run<1>("256 mads after one 4-byte nop (offset 4)");
        run<2>("256 mads after two 4-byte nops (aligned)");
        run<3>("16 mads + one 4-byte v_mov, repeated");
        run<4>("16 mads + one 8-byte v_mov_e64, repeated");
        run<5>("16 mads + s_waitcnt + s_nop, repeated");
        run<6>("16 mads + s_waitcnt, repeated");
    }
    return 0;
}
