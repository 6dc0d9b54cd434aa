// Micro-benchmark: is every CU equally fast for a given instruction class?  One 192-thread workgroup per launch slot (3 waves, like k_tail), many
// workgroups spread over the chip, each records its HW_ID and the s_memtime ticks of a fixed loop of: (0) dependent v_mad_i64_i32, (1) v_mov_b32_dpp
// row_shr, (2) LDS reads + writes, (3) s_barrier rounds, (4) a mix shaped like the row engine's Fp12 product.  Prints min / median / max ticks over the
// workgroups and the slowest CUs.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_cu.hip -o tools/ubench_cu.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
template <int KIND>
__global__ void __launch_bounds__(192) k(uint32_t* out, uint64_t* cyc, uint32_t* hw, int iters) {
    __shared__ uint32_t lds[4096];
    uint32_t t = threadIdx.x;
    uint32_t a = (t * 2654435761u + 1) & 0x0fffffffu, b = (a ^ 0x9e3779b9u) & 0x0fffffffu;
    int64_t x0 = a;
    uint32_t v = a, w = b;
    for (int i = t; i < 4096; i += 192) lds[i] = i * 7;
    __syncthreads();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(R16("v_mad_i64_i32 %0, vcc, %1, %2, %0\n") : "+v"(x0) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 1) asm volatile(R16("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(v), "+v"(w));
        else if (KIND == 2) { R16(v = lds[(v + t) & 4095]; lds[(t * 14 + i) & 4095] = v + w;) }
        else if (KIND == 3) { R4(__syncthreads(); v += w;) }
        else if (KIND == 5) asm volatile(R16(R16(R4("v_mad_i64_i32 %0, vcc, %1, %2, %0\n"))) : "+v"(x0) : "v"(a), "v"(b) : "vcc");      // 1024 instructions = 8 KB of straight-line code per iteration
        else if (KIND == 6) asm volatile(R16(R16(R16("v_mad_i64_i32 %0, vcc, %1, %2, %0\n"))) : "+v"(x0) : "v"(a), "v"(b) : "vcc");     // 4096 instructions = 32 KB
        else {
            R4(v = lds[(v + t) & 4095];)
            asm volatile(R16("v_mad_i64_i32 %0, vcc, %1, %2, %0\n") : "+v"(x0) : "v"(a), "v"(b) : "vcc");
            __syncthreads();
            asm volatile(R4("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mov_b32_dpp %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(v), "+v"(w));
            if ((t & 15) == 15) { asm volatile(R16("v_mad_i64_i32 %0, vcc, %1, %2, %0\n") : "+v"(x0) : "v"(a), "v"(b) : "vcc"); lds[t] = (uint32_t)x0; }
            __syncthreads();
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 192 + t] = (uint32_t)x0 + v + w;
    if (t == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        cyc[blockIdx.x] = t1 - t0; hw[blockIdx.x] = (id & 0xffff) | (xcc << 16);
    }
}
template <int KIND>
void run(const char* name, int blocks, int iters) {
    uint32_t* out; uint64_t* cyc; uint32_t* hw;
    (void)hipMalloc(&out, (size_t)blocks * 192 * 4); (void)hipMalloc(&cyc, blocks * 8); (void)hipMalloc(&hw, blocks * 4);
    k<KIND><<<blocks, 192>>>(out, cyc, hw, iters / 10);
    (void)hipDeviceSynchronize();
    k<KIND><<<blocks, 192>>>(out, cyc, hw, iters);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> c(blocks); std::vector<uint32_t> h(blocks);
    (void)hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h.data(), hw, blocks * 4, hipMemcpyDeviceToHost);
    std::vector<int> idx(blocks); for (int i = 0; i < blocks; i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return c[x] < c[y]; });
    printf("%-34s blocks %4d: ticks min %8llu  median %8llu  max %8llu  (max/min %.2f)  slowest:", name, blocks, (unsigned long long)c[idx[0]], (unsigned long long)c[idx[blocks / 2]],
           (unsigned long long)c[idx[blocks - 1]], (double)c[idx[blocks - 1]] / c[idx[0]]);
    for (int j = 0; j < 4; j++) { int i = idx[blocks - 1 - j]; printf(" xcc%u se%u cu%u(%llu)", (h[i] >> 16) & 15, (h[i] >> 13) & 7, (h[i] >> 8) & 15, (unsigned long long)c[i]); }
    printf("  fastest:");
    for (int j = 0; j < 3; j++) { int i = idx[j]; printf(" xcc%u se%u cu%u", (h[i] >> 16) & 15, (h[i] >> 13) & 7, (h[i] >> 8) & 15); }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(hw);
}
// one 3-wave workgroup ALONE on the chip, launch after launch (each lands on another CU): the condition of the serial tail
template <int KIND>
void alone(const char* name, int iters) {
    uint32_t* out; uint64_t* cyc; uint32_t* hw;
    (void)hipMalloc(&out, 192 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&hw, 4);
    printf("%-34s alone:", name);
    for (int rep = 0; rep < 24; rep++) {
        k<KIND><<<1, 192>>>(out, cyc, hw, iters);
        (void)hipDeviceSynchronize();
        uint64_t c; uint32_t h;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&h, hw, 4, hipMemcpyDeviceToHost);
        printf(" %llu(x%u.s%u.c%u)", (unsigned long long)c / 1000, (h >> 16) & 15, (h >> 13) & 7, (h >> 8) & 15);
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(hw);
}
int main() {
    alone<0>("v_mad_i64_i32 dependent", 4000);
    alone<1>("v_mov_b32_dpp row_shr", 4000);
    alone<2>("LDS read + write", 2000);
    alone<3>("s_barrier rounds (3 waves)", 4000);
    alone<4>("row-engine-shaped mix", 2000);
    alone<5>("8 KB straight-line mad loop", 200);
    alone<6>("32 KB straight-line mad loop", 50);
    for (int blocks : {256, 32}) {
        run<0>("v_mad_i64_i32 dependent", blocks, 4000);
        run<1>("v_mov_b32_dpp row_shr", blocks, 4000);
        run<2>("LDS read + write", blocks, 2000);
        run<3>("s_barrier rounds (3 waves)", blocks, 4000);
        run<4>("row-engine-shaped mix", blocks, 2000);
    }
    return 0;
}
