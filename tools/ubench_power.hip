// Micro-benchmark: does the clock the chip holds under a v_mad_i64_i32 stream depend on the OPERAND VALUES (the chip lowers its clock
// under load, MI355X_MICROARCH.md "DVFS give-back")?  One wave per SIMD on every SIMD, an aligned stream of dependent multiply-adds,
// ~80 ms per case; prints wall ns per multiply-add per SIMD and the in-kernel clock (s_memtime ticks / s_memrealtime at 100 MHz).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_power.hip -o tools/ubench_power.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
__global__ void __launch_bounds__(64) k(uint32_t* out, uint64_t* stamps, int iters, uint32_t amask, uint32_t bmask, int mix) {
    uint32_t a = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u) & amask, b = ((a ^ 0x9e3779b9u) * 2246822519u) & bmask;
    uint32_t a2 = (a * 3u + 7u) & amask, b2 = (b * 5u + 11u) & bmask, a3 = (a * 7u + 1u) & amask, b3 = (b * 9u + 3u) & bmask;
    int64_t x0 = a, x1 = b;
    uint32_t y0 = a, y1 = b;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (mix == 0) {
            REP8(asm volatile(".p2align 3\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %0, vcc, %6, %7, %0\n v_mad_i64_i32 %1, vcc, %2, %5, %1\n"
                              " v_mad_i64_i32 %0, vcc, %4, %7, %0\n v_mad_i64_i32 %1, vcc, %6, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %7, %0\n v_mad_i64_i32 %1, vcc, %4, %3, %1"
                              : "+v"(x0), "+v"(x1) : "v"(a), "v"(b), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc");)
        } else {          // 6 multiply-adds + 2 plain 32-bit instructions (the ~72 % multiply-add share of the real kernels)
            REP8(asm volatile(".p2align 3\n v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %6, %7, %1\n v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_and_b32_e64 %2, %2, %4\n"
                              " v_mad_i64_i32 %1, vcc, %4, %7, %1\n v_mad_i64_i32 %0, vcc, %6, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %5, %1\n v_add_u32_e64 %3, %3, %5"
                              : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1) : "v"(a), "v"(b), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc");)
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x0 + x1) + y0 + y1;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
static void run(const char* name, uint32_t amask, uint32_t bmask, int mix, int blocks) {
    uint32_t* out; uint64_t* st; int iters = 600000;
    hipMalloc(&out, (size_t)blocks * 64 * 4); hipMalloc(&st, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 64>>>(out, st, iters / 10, amask, bmask, mix);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 64>>>(out, st, iters, amask, bmask, mix);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; hipMemcpy(h, st + 2 * (blocks / 2), 16, hipMemcpyDeviceToHost);
    double ninstr = (double)iters * 64, nmad = mix ? ninstr * 0.75 : ninstr;
    printf("%-44s blocks=%4d  wall %.1f ms  ns/instr/SIMD %.3f  ns/mad/SIMD %.3f  ticks/instr %.3f  in-kernel clock %.3f GHz\n", name, blocks, ms, ms * 1e6 / ninstr,
           ms * 1e6 / nmad, (double)h[0] / ninstr, (double)h[0] / ((double)h[1] * 10.0));
    hipFree(out); hipFree(st);
}
int main() {
    for (int blocks : {1024, 128}) {
        run("mad: random 28 x random 28 bit", 0x0fffffffu, 0x0fffffffu, 0, blocks);
        run("mad: random 32 x random 32 bit", 0xffffffffu, 0xffffffffu, 0, blocks);
        run("mad: random 28 x 8-bit", 0x0fffffffu, 0xffu, 0, blocks);
        run("mad: 8-bit x random 28", 0xffu, 0x0fffffffu, 0, blocks);
        run("mad: zero x zero", 0u, 0u, 0, blocks);
        run("6 mad + 2 alu: random 28 x random 28", 0x0fffffffu, 0x0fffffffu, 1, blocks);
        run("6 mad + 2 alu: zero operands", 0u, 0u, 1, blocks);
    }
    return 0;
}
