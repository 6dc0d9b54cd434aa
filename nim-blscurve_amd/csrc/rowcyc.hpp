// Cyclotomic squaring of the final exponentiation's hard part on row arithmetic (round 6).  The 5 x 63 squarings of the exponentiations by |x| are a
// dependent chain of Fp12 operations; the Fp12 engine (c12.hpp) spends ~2.2 us on each whatever it multiplies (a lane's Fp product, two barriers, the
// recombination).  For a unitary element g = A + B w + C w^2 over Fp4 = Fp2[s], s = w^3 (flat basis: A = c0 + c3 s, B = c1 + c4 s, C = c2 + c5 s) the
// Granger-Scott square is
//     g^2 = (3 A^2 - 2 conj A) + (3 s C^2 + 2 conj B) w + (3 B^2 - 2 conj C) w^2,
// three Fp4 squarings = nine Fp2 squarings = EIGHTEEN Fp products: row r = 2 z + kind of the block forms the product (u + v)(u - v) (kind 0) or u v (kind 1)
// of z = 3 e + part (part 0: x0 = c_e, 1: x1 = c_(e+3), 2: x0 + x1) with row_mul (rowfp.hpp, ~0.44 us), twelve rows then combine six products and the old
// coefficient each (small integer coefficients, 64-bit sums, fp_reduce's quotient from the top limb) - two barriers per squaring, ~0.7 us.
// The functions are written on `rw` with the memory behind a template parameter: the device reads the engine's own LDS registers (an fp is 14 packed words),
// tests/host_emu runs them on emulated lanes against tower.hpp's fp12_sqr.
#pragma once
#include "rowfp.hpp"

namespace bls {

ROW_FN rw row_sel(bool c, const rw& a, const rw& b) { return c ? a : b; }

// fp_reduce's quotient step and one linear carry pass on 64-bit limb sums (|.| < 2^37, value below 64 p): |out| < 0.51 p + a little, limbs 0..12 within
// [-2^9, 2^28 + 2^9], limb 13 signed, lanes 14 and 15 zero
ROW_FN rw row_reduce64(const row_ctx& C, rw64 acc) {
    const rw top = row_bcast<FP_N - 1>(row_lo(acc));
    const rw nq = row_neg(row_sar64(row_add64(row_mad(top, row_splat(10322735), row_zero64()), (int64_t)1 << 39), 40));
    acc = row_mad(nq, C.prot[0], acc);
    return (row_lo(acc) & C.maskv) + row_up1(row_hi28(acc) & C.low13);
}

// the six product coefficients (times three already) of output row o = 2 j + comp over (P[x0][0], P[x0][1], P[x1][0], P[x1][1], P[sum][0], P[sum][1]) of its
// Fp4 element, the coefficient of the old value, and the element: j = 0: 3 R_0 - 2 c0, 3: 3 I_0 + 2 c3, 2: 3 R_1 - 2 c2, 5: 3 I_1 + 2 c5, 4: 3 R_2 - 2 c4,
// 1: 3 xi I_2 + 2 c1, with R = Sq(x0) + xi Sq(x1), I = Sq(x0 + x1) - Sq(x0) - Sq(x1), Sq(z) = (P[z][0], 2 P[z][1])
struct cyc_out_row { int8_t k[6]; int8_t old; int8_t e; };
BLS_HD cyc_out_row cyc_out_of(int o) {
    const int j = o >> 1, comp = o & 1;
    const int e = (j == 0 || j == 3) ? 0 : ((j == 2 || j == 5) ? 1 : 2);
    const int type = (j == 0 || j == 2 || j == 4) ? 0 : (j == 1 ? 2 : 1);          // 0: R, 1: I, 2: xi I
    const int8_t T[6][6] = {{1, 0, 1, -2, 0, 0}, {0, 2, 1, 2, 0, 0},                // R.re, R.im
                            {-1, 0, -1, 0, 1, 0}, {0, -2, 0, -2, 0, 2},             // I.re, I.im
                            {-1, 2, -1, 2, 1, -2}, {-1, -2, -1, -2, 1, 2}};         // (xi I).re, (xi I).im
    cyc_out_row r;
    for (int i = 0; i < 6; i++) r.k[i] = (int8_t)(3 * T[2 * type + comp][i]);
    r.old = (int8_t)(type == 0 ? -2 : 2);
    r.e = (int8_t)e;
    return r;
}

// Phase A, product row r (0 .. 17).  mem.coef(j, comp) -> coefficient j (flat basis), component comp of the value being squared, along the row.
template <class Mem>
ROW_FN rw cyc_product_row(const row_ctx& C, const Mem& mem, int r) {
    const int z = r >> 1, kind = r & 1, e = z / 3, part = z % 3;
    const rw a0 = mem.coef(e, 0), a1 = mem.coef(e, 1), b0 = mem.coef(e + 3, 0), b1 = mem.coef(e + 3, 1);
    const rw zero = row_zero();
    const rw u = row_sel(part == 1, zero, a0) + row_sel(part == 0, zero, b0);
    const rw v = row_sel(part == 1, zero, a1) + row_sel(part == 0, zero, b1);
    const rw A = row_norm(C, row_sel(kind == 0, u + v, u)), B = row_norm(C, row_sel(kind == 0, u - v, v));
    return row_mul(C, A, B);
}
// Phase B, output row o (0 .. 11): the new coefficient (o >> 1), component (o & 1).  mem.prod(r) -> product r of phase A.
template <class Mem>
ROW_FN rw cyc_output_row(const row_ctx& C, const Mem& mem, int o, const cyc_out_row& t) {
    rw64 acc = row_mad(mem.coef(o >> 1, o & 1), row_splat(t.old), row_zero64());
    for (int i = 0; i < 6; i++) acc = row_mad(mem.prod(6 * t.e + i), row_splat(t.k[i]), acc);       // products of element e: rows 6 e .. 6 e + 5 in (x0, x1, sum) x (kind) order
    return row_reduce64(C, acc);
}

}  // namespace bls
