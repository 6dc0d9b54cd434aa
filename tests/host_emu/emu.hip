// Host-side execution of the product's __host__ __device__ arithmetic, for CPU-only tests in the
// build container (no GPU there).  TEST INFRASTRUCTURE: never linked into the product library.
#include "fp.hpp"
#include "tower.hpp"
#include "curve.hpp"
#include "h2c.hpp"
#include "pairing.hpp"
#include "deser.hpp"
#include "c12.hpp"
#include "teamvm.hpp"
#define BLS_ROW_EMU 1                    // rowfp.hpp on 64 emulated lanes
#include "rowfp.hpp"
#include "rowvm.hpp"
using namespace bls;
static rw emu_row_of4(const uint8_t* p192) { return row_pick(row_from_fp(fp_load_le(p192)), row_from_fp(fp_load_le(p192 + 48)), row_from_fp(fp_load_le(p192 + 96)), row_from_fp(fp_load_le(p192 + 144))); }
static void emu_row_to4(uint8_t* r192, const rw& x) {
    fp_store_le(r192, fp_reduce(row_to_fp(row_from<0>(x)))); fp_store_le(r192 + 48, fp_reduce(row_to_fp(row_from<1>(x))));
    fp_store_le(r192 + 96, fp_reduce(row_to_fp(row_from<2>(x)))); fp_store_le(r192 + 144, fp_reduce(row_to_fp(row_from<3>(x))));
}
static row_g1 emu_row_g1(const uint8_t* p144) { g1_jac p = g1_jac_load(p144); return row_g1{row_from_fp(p.x), row_from_fp(p.y), row_from_fp(p.z)}; }
static void emu_row_g1_out(uint8_t* o144, const row_g1& r) { g1_jac_store(o144, g1_jac{fp_reduce(row_to_fp(r.x)), fp_reduce(row_to_fp(r.y)), fp_reduce(row_to_fp(r.z))}); }
#define TVM_TABLE static
#include "_build/teamvm_tables.inc"      // written by build.sh (nim-blscurve_amd/tools/teamvm.py): the tables the library embeds
extern "C" {
// The limb-parallel row arithmetic of the MSM's tail (csrc/rowfp.hpp) on 64 emulated lanes, every bound asserted.
// four products at once, one per row
void emu_row_mul(const uint8_t* a192, const uint8_t* b192, uint8_t* r192, int a_twice) {
    const row_ctx C = row_ctx_make();
    rw a = emu_row_of4(a192), b = emu_row_of4(b192);
    if (a_twice) a = a + a;                      // the widest operand the formulas pass in: a doubled product
    emu_row_to4(r192, row_mul(C, a, b));
}
void emu_row_pow(const uint8_t* a192, uint8_t* r192) {                // a^((p-3)/4) of four values, one per row (k_hash_one's square-root chains)
    const row_ctx C = row_ctx_make();
    row_tab_array T;
    emu_row_to4(r192, row_pow_sched(C, emu_row_of4(a192), k::SW_PM3D4, k::SW_PM3D4_LEN, T));
}
void emu_row_dbl(const uint8_t* p144, uint8_t* o144, int times) {
    const row_ctx C = row_ctx_make();
    row_g1 p = emu_row_g1(p144);
    for (int i = 0; i < times; i++) p = row_dbl(C, p);
    emu_row_g1_out(o144, p);
}
void emu_row_add(const uint8_t* p144, const uint8_t* q144, uint8_t* o144, int dbl_first) {
    const row_ctx C = row_ctx_make();
    row_g1 p = emu_row_g1(p144), q = emu_row_g1(q144);
    for (int i = 0; i < dbl_first; i++) p = row_dbl(C, p);         // operands as a doubling leaves them (the Horner walk's)
    emu_row_g1_out(o144, row_add(C, p, q));
}
// The lane-team engine (csrc/teamvm.hpp) on sixteen emulated lanes, under the bounds tracker: the programs the library embeds, round by round.
// q0 | q1 (2 x 288 B, Jacobian) -> clear_cofactor(q0 + q1) as the engine leaves it (288 B)
void emu_tvm_clear(const uint8_t* q576, uint8_t* out288) {
    static fp S[TVM_CLEAR_SLOTS];
    for (auto& v : S) v = fp_zero();
    g2_jac q0 = g2_jac_load(q576), q1 = g2_jac_load(q576 + 288);
    const fp2* in[6] = {&q0.x, &q0.y, &q0.z, &q1.x, &q1.y, &q1.z};
    for (int j = 0; j < 6; j++) { S[TVM_CLEAR_X + 2 * j] = fp_reduce(in[j]->c0); S[TVM_CLEAR_X + 2 * j + 1] = fp_reduce(in[j]->c1); }
    const fp2 cx = fp2_from_const(k::PSI_CX), cy = fp2_from_const(k::PSI_CY);
    S[TVM_CLEAR_CX] = fp_reduce(cx.c0); S[TVM_CLEAR_CX + 1] = fp_reduce(cx.c1); S[TVM_CLEAR_CY] = fp_reduce(cy.c0); S[TVM_CLEAR_CY + 1] = fp_reduce(cy.c1);
    tvm_run_host(S, TVM_CLEAR_DESC, TVM_CLEAR_SEQ, TVM_CLEAR_NSEQ, [](uint32_t, uint32_t, const fp&) {});
    g2_jac_store(out288, g2_jac{fp2{S[TVM_CLEAR_X], S[TVM_CLEAR_X + 1]}, fp2{S[TVM_CLEAR_Y], S[TVM_CLEAR_Y + 1]}, fp2{S[TVM_CLEAR_Z], S[TVM_CLEAR_Z + 1]}});
}
// the 68 lines of one pair from the engine's walk against pairing.hpp's miller_lines: 1 = every coefficient equal mod p
int emu_tvm_lines_equal(const uint8_t* p144, const uint8_t* q288) {
    g1_jac pj = g1_jac_load(p144);
    g2_jac qj = g2_jac_load(q288);
    line_t ref[N_LINES];
    miller_lines(pj, qj, [&](int s, const line_t& l) { ref[s] = l; });
    static fp S[TVM_LINES_SLOTS];
    for (auto& v : S) v = fp_zero();
    S[TVM_LINES_PX] = fp_reduce(pj.x); S[TVM_LINES_PY] = fp_reduce(pj.y); S[TVM_LINES_PZ] = fp_reduce(pj.z);
    const fp2* in[3] = {&qj.x, &qj.y, &qj.z};
    for (int j = 0; j < 3; j++) { S[TVM_LINES_QX + 2 * j] = fp_reduce(in[j]->c0); S[TVM_LINES_QX + 2 * j + 1] = fp_reduce(in[j]->c1); }
    int same = 1, seen = 0;
    tvm_run_host(S, TVM_LINES_DESC, TVM_LINES_SEQ, TVM_LINES_NSEQ, [&](uint32_t step, uint32_t plane, const fp& v) {
        const fp2* l[3] = {&ref[step].l0, &ref[step].l1, &ref[step].l2};
        same &= fp_eq(v, (plane & 1) ? l[plane >> 1]->c1 : l[plane >> 1]->c0) ? 1 : 0;
        seen++;
    });
    return same && seen == N_LINES * 6;
}
// the same two programs on the ROW executor (csrc/rowvm.hpp): sixteen rows = four emulated waves per round
void emu_rvm_clear(const uint8_t* q576, uint8_t* out288) {
    static fp S[TVM_CLEAR_SLOTS];
    for (auto& v : S) v = fp_zero();
    g2_jac q0 = g2_jac_load(q576), q1 = g2_jac_load(q576 + 288);
    const fp2* in[6] = {&q0.x, &q0.y, &q0.z, &q1.x, &q1.y, &q1.z};
    for (int j = 0; j < 6; j++) { S[TVM_CLEAR_X + 2 * j] = fp_reduce(in[j]->c0); S[TVM_CLEAR_X + 2 * j + 1] = fp_reduce(in[j]->c1); }
    const fp2 cx = fp2_from_const(k::PSI_CX), cy = fp2_from_const(k::PSI_CY);
    S[TVM_CLEAR_CX] = fp_reduce(cx.c0); S[TVM_CLEAR_CX + 1] = fp_reduce(cx.c1); S[TVM_CLEAR_CY] = fp_reduce(cy.c0); S[TVM_CLEAR_CY + 1] = fp_reduce(cy.c1);
    rvm_run_host(S, TVM_CLEAR_DESC, TVM_CLEAR_SEQ, TVM_CLEAR_NSEQ, [](uint32_t, uint32_t, const fp&) {});
    g2_jac_store(out288, g2_jac{fp2{S[TVM_CLEAR_X], S[TVM_CLEAR_X + 1]}, fp2{S[TVM_CLEAR_Y], S[TVM_CLEAR_Y + 1]}, fp2{S[TVM_CLEAR_Z], S[TVM_CLEAR_Z + 1]}});
}
// the 68 lines of one pair from the engine's walk against pairing.hpp's miller_lines: 1 = every coefficient equal mod p
int emu_rvm_lines_equal(const uint8_t* p144, const uint8_t* q288) {
    g1_jac pj = g1_jac_load(p144);
    g2_jac qj = g2_jac_load(q288);
    line_t ref[N_LINES];
    miller_lines(pj, qj, [&](int s, const line_t& l) { ref[s] = l; });
    static fp S[TVM_LINES_SLOTS];
    for (auto& v : S) v = fp_zero();
    S[TVM_LINES_PX] = fp_reduce(pj.x); S[TVM_LINES_PY] = fp_reduce(pj.y); S[TVM_LINES_PZ] = fp_reduce(pj.z);
    const fp2* in[3] = {&qj.x, &qj.y, &qj.z};
    for (int j = 0; j < 3; j++) { S[TVM_LINES_QX + 2 * j] = fp_reduce(in[j]->c0); S[TVM_LINES_QX + 2 * j + 1] = fp_reduce(in[j]->c1); }
    int same = 1, seen = 0;
    rvm_run_host(S, TVM_LINES_DESC, TVM_LINES_SEQ, TVM_LINES_NSEQ, [&](uint32_t step, uint32_t plane, const fp& v) {
        const fp2* l[3] = {&ref[step].l0, &ref[step].l1, &ref[step].l2};
        same &= fp_eq(v, (plane & 1) ? l[plane >> 1]->c1 : l[plane >> 1]->c0) ? 1 : 0;
        seen++;
    });
    return same && seen == N_LINES * 6;
}
void emu_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { fp_store_le(r, fp_mul(fp_load_le(a), fp_load_le(b))); }
void emu_fp_add(const uint8_t* a, const uint8_t* b, uint8_t* r) { fp_store_le(r, fp_add(fp_load_le(a), fp_load_le(b))); }
void emu_fp_sub(const uint8_t* a, const uint8_t* b, uint8_t* r) { fp_store_le(r, fp_sub(fp_load_le(a), fp_load_le(b))); }
void emu_fp_inv(const uint8_t* a, uint8_t* r) { fp_store_le(r, fp_inv(fp_load_le(a))); }
void emu_fp_inv_fermat(const uint8_t* a, uint8_t* r) { fp_store_le(r, fp_inv_fermat(fp_load_le(a))); }
void emu_fp2_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { fp2_store_le(r, fp2_mul(fp2_load_le(a), fp2_load_le(b))); }
void emu_fp2_sqr(const uint8_t* a, uint8_t* r) { fp2_store_le(r, fp2_sqr(fp2_load_le(a))); }
void emu_fp2_inv(const uint8_t* a, uint8_t* r) { fp2_store_le(r, fp2_inv(fp2_load_le(a))); }
void emu_sha256(const uint8_t* m, uint32_t n, uint8_t* out) {
    sha256_ctx c; sha256_begin(c); sha256_update(c, m, n); uint32_t d[8]; sha256_end(c, d);
    for (int i = 0; i < 8; i++) { out[4*i] = d[i] >> 24; out[4*i+1] = d[i] >> 16; out[4*i+2] = d[i] >> 8; out[4*i+3] = d[i]; }
}
void emu_hash_to_field(const uint8_t* m, uint32_t n, const uint8_t* dst, uint32_t dn, uint8_t* out192) {
    fp2 u0, u1; hash_to_field_fp2x2(u0, u1, m, n, dst, dn); fp2_store_le(out192, u0); fp2_store_le(out192 + 96, u1);
}
// the 32-byte-message fast path of k_hash_map (constants per DST + 18 compressions); returns 0 when the DST length is outside its range
int emu_hash_to_field_msg32(const uint8_t* m32, const uint8_t* dst, uint32_t dn, uint8_t* out192) {
    xmd32_consts c = xmd32_precompute(dst, dn);
    if (!c.valid) return 0;
    uint32_t mbe[8];
    xmd32_pack(mbe, m32, 8);
    fp2 u0, u1; hash_to_field_fp2x2_msg32(u0, u1, mbe, c); fp2_store_le(out192, u0); fp2_store_le(out192 + 96, u1);
    return 1;
}
// u (96 B) -> Jacobian point on E2' (288 B)
void emu_sswu(const uint8_t* u, uint8_t* out) { g2_jac_store(out, sswu_g2(fp2_load_le(u))); }
void emu_iso3(const uint8_t* in, uint8_t* out) { g2_jac_store(out, iso3_g2(g2_jac_load(in))); }
void emu_hash_to_g2(const uint8_t* m, uint32_t n, const uint8_t* dst, uint32_t dn, uint8_t* out) { g2_jac_store(out, hash_to_g2(m, n, dst, dn)); }
void emu_g1_mul_u64(const uint8_t* p, uint64_t k, uint8_t* out) { g1_jac_store(out, jac_mul_u64(g1_aff_load(p), k)); }
void emu_g1_mul_u64_w4(const uint8_t* p, uint64_t k, uint8_t* out) { g1_jac_store(out, jac_mul_u64_w4(g1_aff_load(p), k)); }
void emu_g2_mul_u64_w4(const uint8_t* p, uint64_t k, uint8_t* out) { g2_jac_store(out, jac_mul_u64_w4(g2_aff_load(p), k)); }
void emu_g2_mul_u64(const uint8_t* p, uint64_t k, uint8_t* out) { g2_jac_store(out, jac_mul_u64(g2_aff_load(p), k)); }
void emu_g1_mul_256(const uint8_t* p, const uint8_t* k32, uint8_t* out) {
    uint32_t kk[8];
    for (int j = 0; j < 8; j++) kk[j] = (uint32_t)k32[4 * j] | ((uint32_t)k32[4 * j + 1] << 8) | ((uint32_t)k32[4 * j + 2] << 16) | ((uint32_t)k32[4 * j + 3] << 24);
    g1_jac_store(out, jac_mul_256(g1_aff_load(p), kk));
}
void emu_g2_mul_256_jac(const uint8_t* p, const uint8_t* k32, uint8_t* out) {
    uint32_t kk[8];
    for (int j = 0; j < 8; j++) kk[j] = (uint32_t)k32[4 * j] | ((uint32_t)k32[4 * j + 1] << 8) | ((uint32_t)k32[4 * j + 2] << 16) | ((uint32_t)k32[4 * j + 3] << 24);
    g2_jac_store(out, jac_mul_256_jac(g2_jac_load(p), kk));
}
// the lane-cooperative formulas with the solo team (every product computed here): must equal the plain ones
void emu_g2_dbl_team(const uint8_t* p, uint8_t* out) { g2_jac_store(out, jac_dbl_team(g2_jac_load(p), team_solo{})); }
void emu_g2_dbl(const uint8_t* p, uint8_t* out) { g2_jac_store(out, jac_dbl(g2_jac_load(p))); }
void emu_g2_add_team(const uint8_t* a, const uint8_t* b, uint8_t* out) { g2_jac_store(out, jac_add_team(g2_jac_load(a), g2_jac_load(b), team_solo{})); }
void emu_g1_dbl_team(const uint8_t* p, uint8_t* out) { g1_jac_store(out, jac_dbl_team(g1_jac_load(p), team_solo{})); }
void emu_g1_dbl(const uint8_t* p, uint8_t* out) { g1_jac_store(out, jac_dbl(g1_jac_load(p))); }
void emu_g1_add_team(const uint8_t* a, const uint8_t* b, uint8_t* out) { g1_jac_store(out, jac_add_team(g1_jac_load(a), g1_jac_load(b), team_solo{})); }
// sum of n affine points (96 / 192 bytes each, all-zero = infinity) with the extended-Jacobian mixed addition of the Pippenger buckets
void emu_g1_sum_xyzz(const uint8_t* pts, uint32_t n, uint8_t* out) {
    xyzz<fp> acc = xyzz_inf<fp>();
    for (uint32_t i = 0; i < n; i++) acc = xyzz_add_aff(acc, g1_aff_load(pts + 96 * i));
    g1_jac_store(out, jac_from_xyzz(acc));
}
void emu_g2_sum_xyzz(const uint8_t* pts, uint32_t n, uint8_t* out) {
    xyzz<fp2> acc = xyzz_inf<fp2>();
    for (uint32_t i = 0; i < n; i++) acc = xyzz_add_aff(acc, g2_aff_load(pts + 192 * i));
    g2_jac_store(out, jac_from_xyzz(acc));
}
void emu_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { g1_jac_store(out, jac_add(g1_jac_load(a), g1_jac_load(b))); }
void emu_g2_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { g2_jac_store(out, jac_add(g2_jac_load(a), g2_jac_load(b))); }
// n pairs of (P Jacobian 144 B, Q Jacobian 288 B) -> final_exp(miller) 576 B
void emu_pairing_product(const uint8_t* ps, const uint8_t* qs, uint32_t n, uint8_t* out, int do_final_exp) {
    fp12 L[N_LINES];
    for (int s = 0; s < N_LINES; s++) L[s] = fp12_one();
    for (uint32_t i = 0; i < n; i++)
        miller_lines(g1_jac_load(ps + 144 * i), g2_jac_load(qs + 288 * i), [&](int s, const line_t& l) { L[s] = fp12_mul_by_line(L[s], l); });
    fp12 f = miller_combine([&](int s) { return fp12_reduce(L[s]); });      // as k_lineprod does before its product tree
    if (do_final_exp) f = final_exp(f);
    fp12_store_le(out, f);
}
// compressed -> blst affine image; returns 1 ok / 0 bad encoding; *inf set for the infinity encoding
int emu_g1_uncompress(const uint8_t* b48, uint8_t* out96, int* inf) {
    g1_aff a; bool i; bool ok = g1_uncompress(a, i, b48); *inf = i;
    fp_store_le(out96, a.x); fp_store_le(out96 + 48, a.y); return ok;
}
int emu_g2_uncompress(const uint8_t* b96, uint8_t* out192, int* inf) {
    g2_aff a; bool i; bool ok = g2_uncompress(a, i, b96); *inf = i;
    fp2_store_le(out192, a.x); fp2_store_le(out192 + 96, a.y); return ok;
}
// 96- / 192-byte forms (blst_pN_deserialize semantics)
int emu_g1_deserialize(const uint8_t* b96, uint8_t* out96, int* inf) {
    g1_aff a; bool i; bool ok = g1_deserialize(a, i, b96); *inf = i;
    fp_store_le(out96, a.x); fp_store_le(out96 + 48, a.y); return ok;
}
int emu_g2_deserialize(const uint8_t* b192, uint8_t* out192, int* inf) {
    g2_aff a; bool i; bool ok = g2_deserialize(a, i, b192); *inf = i;
    fp2_store_le(out192, a.x); fp2_store_le(out192 + 96, a.y); return ok;
}
int emu_deserialize_tuple_ex(const uint8_t* pk, const uint8_t* sig, uint32_t flags) { g1_aff p; g2_aff s; return deserialize_tuple(p, s, pk, sig, flags); }
int emu_g1_in_subgroup(const uint8_t* aff96) { return g1_in_subgroup(g1_aff_load(aff96)); }
int emu_g2_in_subgroup(const uint8_t* aff192) { return g2_in_subgroup(g2_aff_load(aff192)); }
int emu_deserialize_tuple(const uint8_t* pk48, const uint8_t* sig96) { g1_aff p; g2_aff s; return deserialize_tuple(p, s, pk48, sig96, 0); }
void emu_fp12_mul(const uint8_t* a, const uint8_t* b, uint8_t* r) { fp12_store_le(r, fp12_mul(fp12_load_le(a), fp12_load_le(b))); }
void emu_final_exp(const uint8_t* a, uint8_t* r) { fp12_store_le(r, final_exp(fp12_load_le(a))); }
// Multiply-add census of ONE tuple through each stage of the one-lane-per-item pipeline (the formulas the kernels run):
//   0 k_hash_map (hash_to_field + 2 x SSWU + isogeny)   1 k_hash_clear (add + cofactor clearing)   2 k_pkmul (load + [r]PK)
//   3 signature side (convert + 8 bucket additions)     4 k_lines (68 lines)                       5 k_lineprod (68 sparse products)
unsigned long long emu_mad_census(int stage, const uint8_t* set320, uint64_t r) {
#if defined(BLS_TRACK_BOUNDS)
    const uint8_t dst[] = "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_";
    fp2 u0, u1;
    hash_to_field_fp2x2(u0, u1, set320 + 96, 32, dst, sizeof(dst) - 1);
    g2_jac q0 = iso3_g2(sswu_g2(u0)), q1 = iso3_g2(sswu_g2(u1));
    g2_jac h = clear_cofactor_g2(jac_add(q0, q1));
    g1_aff pk = g1_aff_load(set320);
    g2_aff sg = g2_aff_load(set320 + 128);
    g1_jac rp = jac_mul_u64_w4(pk, r);
    line_t L[N_LINES];
    miller_lines(rp, h, [&](int s, const line_t& l) { L[s] = l; });
    g_mad_count = 0;
    switch (stage) {
        case 0: hash_to_field_fp2x2(u0, u1, set320 + 96, 32, dst, sizeof(dst) - 1); (void)iso3_g2(sswu_g2(u0)); (void)iso3_g2(sswu_g2(u1)); break;
        case 1: (void)clear_cofactor_g2(jac_add(q0, q1)); break;
        case 2: (void)jac_mul_u64_w4(g1_aff_load(set320), r); break;
        case 3: { g2_aff s2 = g2_aff_load(set320 + 128); xyzz<fp2> acc = xyzz_dbl_aff(sg); g_mad_count = 0; for (int i = 0; i < 8; i++) acc = xyzz_add_aff(acc, s2); } break;      // 8 mixed additions (one per 8-bit window) onto a bucket that is neither infinity nor the point itself
        case 4: miller_lines(rp, h, [&](int, const line_t&) {}); break;
        case 5: { fp12 f = fp12_from_line(L[0]); for (int s = 0; s < N_LINES; s++) f = fp12_mul_by_line(f, L[s]); } break;
    }
    return g_mad_count;
#else
    (void)stage; (void)set320; (void)r;
    return 0;
#endif
}
// the lane-cooperative Fp12 engine of k_tail, run item by item (what the lanes do between barriers): schoolbook products (c12s_product), the limb sums of
// every output coefficient (c12s_limb_sum) and the row reduction (c12_row_reduce_ref: what the device does with DPP row shifts); one operand with negated limbs
void emu_c12_rowphase(const uint8_t* a576, const uint8_t* b576, int sqr, uint8_t* out576) {
    fp12 fa = fp12_load_le(a576), fb = fp12_load_le(b576);
    const fp2* ta[6] = {&fa.c0.a0, &fa.c0.a1, &fa.c0.a2, &fa.c1.a0, &fa.c1.a1, &fa.c1.a2};
    const fp2* tb[6] = {&fb.c0.a0, &fb.c0.a1, &fb.c0.a2, &fb.c1.a0, &fb.c1.a1, &fb.c1.a2};
    fp2 A[6], B[6], D[6];
    for (int t = 0; t < 6; t++) { A[c12_flat_of_tower(t)] = fp2_reduce(*ta[t]); B[c12_flat_of_tower(t)] = fp2_neg(fp2_reduce(fp2_neg(*tb[t]))); }
    static c12_work W;
    for (int q = 0; q < (sqr ? 84 : 144); q++) W.prod[q] = c12s_product(A, sqr ? A : B, q, sqr != 0);
    for (int c = 0; c < 12; c++) {
        int64_t s[16] = {0};
        for (int l = 0; l < FP_N; l++) s[l] = c12s_limb_sum(W, c, l, sqr != 0);
        int32_t o[FP_N];
        c12_row_reduce_ref(s, o);
        fp v;
        for (int l = 0; l < FP_N; l++) v.l[l] = (uint32_t)o[l];
        BLS_SET_VB(v, 1); BLS_SET_LB(v, 1);
        if (c & 1) D[c >> 1].c1 = v; else D[c >> 1].c0 = v;
    }
    fp12 r;
    fp2* tr[6] = {&r.c0.a0, &r.c0.a1, &r.c0.a2, &r.c1.a0, &r.c1.a1, &r.c1.a2};
    for (int t = 0; t < 6; t++) *tr[t] = D[c12_flat_of_tower(t)];
    fp12_store_le(out576, r);
}
// n Granger-Scott squarings of a UNITARY Fp12 value on row arithmetic (csrc/rowcyc.hpp: what c12_cyc_sqr_rows does between its barriers): eighteen
// product rows, twelve output rows, every bound asserted
struct emu_cyc_mem {
    const fp2* A;           // the value, flat basis
    const rw* P;            // the eighteen products
    rw coef(int j, int comp) const { return row_from_fp(comp ? A[j].c1 : A[j].c0); }
    rw prod(int r) const { return P[r]; }
};
void emu_cyc_sqr(const uint8_t* a576, int n, uint8_t* out576) {
    fp12 fa = fp12_load_le(a576);
    const fp2* ta[6] = {&fa.c0.a0, &fa.c0.a1, &fa.c0.a2, &fa.c1.a0, &fa.c1.a1, &fa.c1.a2};
    fp2 A[6];
    for (int t = 0; t < 6; t++) A[c12_flat_of_tower(t)] = fp2_reduce(*ta[t]);
    const row_ctx C = row_ctx_make();
    for (int it = 0; it < n; it++) {
        rw P[18];
        for (int r = 0; r < 18; r++) P[r] = cyc_product_row(C, emu_cyc_mem{A, P}, r);
        fp2 D[6];
        for (int o = 0; o < 12; o++) {
            fp v = row_to_fp(cyc_output_row(C, emu_cyc_mem{A, P}, o, cyc_out_of(o)));
            BLS_SET_VB(v, 1); BLS_SET_LB(v, 1);
            if (o & 1) D[o >> 1].c1 = v; else D[o >> 1].c0 = v;
        }
        for (int j = 0; j < 6; j++) A[j] = D[j];
    }
    fp12 r;
    fp2* tr[6] = {&r.c0.a0, &r.c0.a1, &r.c0.a2, &r.c1.a0, &r.c1.a1, &r.c1.a2};
    for (int t = 0; t < 6; t++) *tr[t] = A[c12_flat_of_tower(t)];
    fp12_store_le(out576, r);
}
void emu_c12_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { emu_c12_rowphase(a, b, 0, out); }
void emu_c12_sqr(const uint8_t* a, uint8_t* out) { emu_c12_rowphase(a, a, 1, out); }
}
