// GPU check of the G1 lane teams (team_quad_fp<2>, <4>: kernels.hip) against the single-lane formulas: every team adds / doubles its own pair of
// points ([a]G, [b]G, also P + P, P + (-P), infinity operands) and compares projectively.  Prints the number of mismatching teams (0 = pass).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 --gpu-max-threads-per-block=64 -mllvm -amdgpu-dpp-combine=false -I nim-blscurve_amd/csrc -I include tools/test_team.hip -o tools/test_team.bin
// (without -amdgpu-dpp-combine=false the general-position addition fails in the even lanes: see tools/test_dpp.hip)
#include "../nim-blscurve_amd/csrc/kernels.hip"
__device__ bool jac_same(const g1_jac& a, const g1_jac& b) {
    bool ai = jac_is_inf(a), bi = jac_is_inf(b);
    if (ai || bi) return ai && bi;
    fp za = fp_sqr(a.z), zb = fp_sqr(b.z);
    return fp_eq(fp_mul(a.x, zb), fp_mul(b.x, za)) && fp_eq(fp_mul(a.y, fp_mul(zb, b.z)), fp_mul(b.y, fp_mul(za, a.z)));
}
template <int T>
__global__ void __launch_bounds__(WAVE) k_test(uint32_t* bad) {
    uint32_t team_id = (blockIdx.x * WAVE + threadIdx.x) / T;
    const team_quad_fp<T> team{threadIdx.x & (T - 1)};
    g1_aff g{fp_from_const(k::G1_X), fp_from_const(k::G1_Y)};
    g1_jac P = jac_mul_u64(g, 3 + 7 * team_id), Q = jac_mul_u64(g, 1000003 + 11 * team_id);
    uint32_t kind = team_id % 6;
    if (kind == 1) Q = P;
    if (kind == 2) Q = g1_jac{P.x, fp_neg(P.y), P.z};
    if (kind == 3) P = jac_inf<fp>();
    if (kind == 4) Q = jac_inf<fp>();
    if (kind == 5) { P = jac_inf<fp>(); Q = P; }
    g1_jac r1 = jac_add_team(P, Q, team), r2 = jac_add_body(P, Q);
    g1_jac d1 = jac_dbl_team(Q, team), d2 = jac_dbl(Q);
    uint32_t f = (jac_same(r1, r2) ? 0u : 1u) | (jac_same(d1, d2) ? 0u : 2u);
    if (f) atomicOr(bad + (f & 1 ? kind : 6 + kind), 1u << (threadIdx.x & (T - 1)));
}
template <int T>
__global__ void __launch_bounds__(WAVE) k_prim(uint32_t* bad) {
    const team_quad_fp<T> team{threadIdx.x & (T - 1)};
    uint32_t team_id = (blockIdx.x * WAVE + threadIdx.x) / T;
    g1_aff g{fp_from_const(k::G1_X), fp_from_const(k::G1_Y)};
    g1_jac P = jac_mul_u64(g, 3 + 7 * team_id), Q = jac_mul_u64(g, 1000003 + 11 * team_id);
    fp r[4], e[4];
    uint32_t lane = threadIdx.x & (T - 1);
    team.mul4(r[0], r[1], r[2], r[3], P.x, Q.x, P.y, Q.y, P.z, Q.z, P.x, Q.z);
    e[0] = fp_mul(P.x, Q.x); e[1] = fp_mul(P.y, Q.y); e[2] = fp_mul(P.z, Q.z); e[3] = fp_mul(P.x, Q.z);
    for (int i = 0; i < 4; i++) if (!fp_eq(r[i], e[i])) atomicOr(bad + i, 1u << lane);
    team.mul2(r[0], r[1], P.x, Q.x, P.y, Q.y);
    for (int i = 0; i < 2; i++) if (!fp_eq(r[i], e[i])) atomicOr(bad + 4 + i, 1u << lane);
    team.sqr2(r[0], r[1], P.x, Q.y);
    e[0] = fp_sqr(P.x); e[1] = fp_sqr(Q.y);
    for (int i = 0; i < 2; i++) if (!fp_eq(r[i], e[i])) atomicOr(bad + 6 + i, 1u << lane);
    team.sqr3(r[0], r[1], r[2], P.x, Q.y, P.z);
    e[2] = fp_sqr(P.z);
    for (int i = 0; i < 3; i++) if (!fp_eq(r[i], e[i])) atomicOr(bad + 8 + i, 1u << lane);
    team.mul3(r[0], r[1], r[2], P.x, Q.x, P.y, Q.y, P.z, Q.z);
    e[0] = fp_mul(P.x, Q.x); e[1] = fp_mul(P.y, Q.y); e[2] = fp_mul(P.z, Q.z);
    for (int i = 0; i < 3; i++) if (!fp_eq(r[i], e[i])) atomicOr(bad + 11 + i, 1u << lane);
}
template <int T>
__global__ void __launch_bounds__(WAVE) k_coord(uint32_t* bad) {
    const team_quad_fp<T> team{threadIdx.x & (T - 1)};
    uint32_t team_id = (blockIdx.x * WAVE + threadIdx.x) / T, lane = threadIdx.x & (T - 1);
    g1_aff g{fp_from_const(k::G1_X), fp_from_const(k::G1_Y)};
    g1_jac P = jac_mul_u64(g, 3 + 7 * team_id), Q = jac_mul_u64(g, 1000003 + 11 * team_id);
    g1_jac a = jac_add_team(P, Q, team), b = jac_add_team(P, Q, team_solo{});
    if (!fp_eq(a.x, b.x)) atomicOr(bad + 0, 1u << lane);
    if (!fp_eq(a.y, b.y)) atomicOr(bad + 1, 1u << lane);
    if (!fp_eq(a.z, b.z)) atomicOr(bad + 2, 1u << lane);
}
int main() {
    uint32_t* bad; (void)hipMalloc(&bad, 64); 
    for (int T : {2, 4}) {
        (void)hipMemset(bad, 0, 64);
        if (T == 2) k_test<2><<<4, WAVE>>>(bad); else k_test<4><<<4, WAVE>>>(bad);
        uint32_t h[16]; (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
        printf("T=%d add-mismatch lanes by kind:", T); for (int i = 0; i < 6; i++) printf(" %x", h[i]);
        printf("  dbl-mismatch:"); for (int i = 6; i < 12; i++) printf(" %x", h[i]);
        printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
    }
    for (int T : {2, 4}) {
        (void)hipMemset(bad, 0, 64);
        if (T == 2) k_prim<2><<<4, WAVE>>>(bad); else k_prim<4><<<4, WAVE>>>(bad);
        uint32_t h[16]; (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
        printf("T=%d primitives, mismatching lanes: mul4", T); for (int i = 0; i < 4; i++) printf(" %x", h[i]);
        printf(" mul2 %x %x sqr2 %x %x sqr3 %x %x %x mul3 %x %x %x\n", h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13]);
    }
    for (int T : {2, 4}) {
        (void)hipMemset(bad, 0, 64);
        if (T == 2) k_coord<2><<<4, WAVE>>>(bad); else k_coord<4><<<4, WAVE>>>(bad);
        uint32_t h[16]; (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
        printf("T=%d add vs solo, mismatching lanes: x %x y %x z %x\n", T, h[0], h[1], h[2]);
    }
    return 0;
}
