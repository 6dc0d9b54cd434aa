// Hardware check (gfx950): which operand of a VOP2 instruction a DPP quad_perm permutes.  Documented: src0.  Measured on MI355X: src0 for v_sub_u32 / v_add_u32,
// but src1 for the "rev" opcodes v_subrev_u32 / v_lshlrev_b32 - LLVM's DPP combine folds a v_mov_b32_dpp into v_subrev_u32_dpp assuming src0, so the library is
// built with -mllvm -amdgpu-dpp-combine=false (build.sh).  Build: hipcc -O3 --offload-arch=gfx950 tools/test_dpp.hip -o tools/test_dpp.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int l = threadIdx.x, a = 1000 + l, x = 7 * l + 3;
    int r1, r2, r3, r4;
    asm volatile("s_nop 4\n v_subrev_u32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r1) : "v"(x), "v"(a));
    asm volatile("s_nop 4\n v_sub_u32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r2) : "v"(x), "v"(a));
    asm volatile("s_nop 4\n v_add_u32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r3) : "v"(x), "v"(a));
    asm volatile("s_nop 4\n v_lshlrev_b32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "=&v"(r4) : "v"(l), "v"(a));
    out[l * 4] = r1; out[l * 4 + 1] = r2; out[l * 4 + 2] = r3; out[l * 4 + 3] = r4;
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 4; l++) {
        int s = (l & ~1) | 1; int x = 7 * l + 3, xs = 7 * s + 3, a = 1000 + l, as = 1000 + s;
        printf("lane %d: subrev %d (S1-dpp(S0)=%d, dpp(S1)-S0=%d)  sub %d (dpp(S0)-S1=%d, S0-dpp(S1)=%d)  add %d (dpp(S0)+S1=%d, S0+dpp(S1)=%d) lshlrev %d (S1<<dpp(S0)=%d, dpp(S1)<<S0=%d)\n", l,
               h[4 * l], a - xs, as - x, h[4 * l + 1], xs - a, x - as, h[4 * l + 2], xs + a, x + as, h[4 * l + 3], a << s, as << l);
    }
}
