// Micro-benchmark: issue cost of v_mad_i64_i32 streams by DEPENDENCY DISTANCE and operand-register pattern, one wave per SIMD,
// 128 workgroups (an eighth of the chip: no clock throttling, one wave per SIMD guaranteed) - cycles per instruction from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_dep.hip -o tools/ubench_dep.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* cyc, int iters) {
    uint32_t a = (threadIdx.x * 2654435761u + 1) & 0x0fffffffu, b = (a ^ 0x9e3779b9u) & 0x0fffffffu;
    uint32_t c = (a * 3 + 1) & 0x0fffffffu, d = (b * 5 + 7) & 0x0fffffffu, e = (a * 7 + 1) & 0x0fffffffu, f = (b * 9 + 7) & 0x0fffffffu;
    int64_t x0 = a, x1 = b, x2 = c, x3 = d;
    uint32_t m = a;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)        // distance 1, same operands
            asm volatile(".p2align 6\n" R64("v_mad_i64_i32 %0, vcc, %4, %5, %0\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 1)   // distance 2
            asm volatile(".p2align 6\n" R16("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 2)   // distance 4
            asm volatile(".p2align 6\n" R16("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 3)   // distance 1, six different operand registers in rotation
            asm volatile(".p2align 6\n" R16("v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %3, %4, %0\n v_mad_i64_i32 %0, vcc, %5, %6, %0\n v_mad_i64_i32 %0, vcc, %1, %4, %0\n")
                         : "+v"(x0) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f) : "vcc");
        else if (KIND == 4)   // the multiplier's column end: 12 chained multiply-adds, m = (lo * N0) & mask, multiply-add with m, 64-bit shift
            asm volatile(".p2align 6\n s_mov_b32 s21, 0xfffffff\n" R4(R4("v_mad_i64_i32 v[20:21], vcc, %1, %2, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %2, %1, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %1, %1, v[20:21]\n")
                                         "v_mul_lo_u32 %0, v20, %2\n v_and_b32_e64 %0, s21, %0\n v_mad_i64_i32 v[20:21], vcc, %0, %1, v[20:21]\n v_ashrrev_i64 v[20:21], 28, v[20:21]\n")
                         : "+v"(m) : "v"(a), "v"(b) : "vcc", "v20", "v21", "s21");
        else if (KIND == 5)   // the same with the three bookkeeping instructions replaced by multiply-adds (what the bookkeeping costs beyond its slots)
            asm volatile(".p2align 6\n" R4(R4("v_mad_i64_i32 v[20:21], vcc, %1, %2, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %2, %1, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %1, %1, v[20:21]\n")
                                         "v_mad_i64_i32 v[20:21], vcc, %1, %2, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %1, %2, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %0, %1, v[20:21]\n v_mad_i64_i32 v[20:21], vcc, %1, %2, v[20:21]\n")
                         : "+v"(m) : "v"(a), "v"(b) : "vcc", "v20", "v21");
        else if (KIND == 6)   // 16 independent 32-bit adds between multiply-add groups: plain ALU cost at one wave
            asm volatile(".p2align 6\n" R16("v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_add_u32_e64 %1, %1, %2\n v_mad_i64_i32 %0, vcc, %3, %2, %0\n v_xor_b32_e64 %1, %1, %3\n")
                         : "+v"(x0), "+v"(m) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 7)   // v_ashrrev_i64 alone, dependent
            asm volatile(".p2align 6\n" R64("v_ashrrev_i64 %0, 1, %0\n") : "+v"(x0) : : "vcc");
        else if (KIND == 8)   // v_mul_lo_u32 alone, dependent
            asm volatile(".p2align 6\n" R64("v_mul_lo_u32 %0, %0, %1\n") : "+v"(m) : "v"(a) : "vcc");
        else if (KIND == 9)   // v_and_b32 alone, dependent
            asm volatile(".p2align 6\n" R64("v_and_b32_e64 %0, %0, %1\n") : "+v"(m) : "v"(a) : "vcc");
        else if (KIND == 10)  // multiply-add reading a SGPR constant (the m * p products)
            asm volatile(".p2align 6\n s_mov_b32 s20, 0xfffaaab\n" R64("v_mad_i64_i32 %0, vcc, %1, s20, %0\n") : "+v"(x0) : "v"(a) : "vcc", "s20");
        else if (KIND == 11)  // v_accvgpr_read / write pairs (the AGPR moves around the calls)
            asm volatile(".p2align 6\n" R16("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a1\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_read_b32 %0, a3\n") : "+v"(m) : : "a0", "a1", "a2", "a3");
        else if (KIND == 12)  // plain moves, independent
            asm volatile(".p2align 6\n" R16("v_mov_b32_e64 %0, %4\n v_mov_b32_e64 %1, %5\n v_mov_b32_e64 %2, %4\n v_mov_b32_e64 %3, %5\n") : "+v"(m), "+v"(c), "+v"(d), "+v"(e) : "v"(a), "v"(b));
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1 + x2 + x3) + m + c + d + e;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int per_iter) {
    uint32_t* out; uint64_t* cyc; int iters = 20000, blocks = 128;
    (void)hipMalloc(&out, (size_t)blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
    k<KIND><<<blocks, 64>>>(out, cyc, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<KIND><<<blocks, 64>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t c0; (void)hipMemcpy(&c0, cyc + blocks / 2, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * per_iter;
    printf("%-72s cycles/instr=%6.3f  wall ns/instr=%6.3f\n", name, (double)c0 / n, ms * 1e6 / n);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("mad, dependent (distance 1), same operands", 64);
    run<1>("mad, two accumulators alternating (distance 2)", 64);
    run<2>("mad, four accumulators (distance 4)", 64);
    run<3>("mad, dependent, six operand registers in rotation", 64);
    run<4>("multiplier column: 12 mads + mul_lo + and + mad + ashr64", 64);
    run<5>("the same 16 slots as multiply-adds only", 64);
    run<6>("mad / add / mad / xor (independent 32-bit ALU between mads)", 64);
    run<7>("v_ashrrev_i64 dependent", 64);
    run<8>("v_mul_lo_u32 dependent", 64);
    run<9>("v_and_b32 dependent", 64);
    run<10>("mad with an SGPR operand, dependent", 64);
    run<11>("v_accvgpr_write / read", 64);
    run<12>("v_mov_b32 independent", 64);
    return 0;
}
