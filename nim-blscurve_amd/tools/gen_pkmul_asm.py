#!/usr/bin/env python3
"""Generates the body of k_pkmul (csrc/kernels.hip) - [r_i] PK_i for a 64-bit blinding scalar and an affine public key per lane, the public-key
side of the batch check (blst_pairing_chk_n_mul_n_aggr_pk_in_g1's scalar multiplication, blst_abi.nim:462-475; core :545-566) - as ONE
hand-allocated gfx950 inline-asm statement.

Algorithm (curve.hpp jac_mul_u64_w4_body's, with a bias instead of a carry chain for the signed digits): k' = r + 0x8888888888888888, digit j =
nibble j of k' minus 8 in [-8, 7] (j < 16), digit 16 = the carry (0 or 1); a table of 1 .. 8 times PK in Jacobian form WITH each entry's Z^2 and
Z^3 (curve.hpp jac_precompute), then per window four doublings (jac_dbl_lazy's G1 form: 3 squares, 2 products, one a b - c d) and one addition of
+-entry (jac_add_pre: 3 squares, 9 products, one a b - c d).
Why assembly: the compiled kernel (0.69 multiply-add share, 16 % of its wave cycles waiting, 1.5 GB of HBM traffic for 16 MB of input) keeps its
table in scratch memory, indexed by a per-lane digit right where it is needed.  Here the multiplier bodies are expanded in place on fixed
registers (VGPRs only: two waves per SIMD as before), the table lives in a global buffer of the context, 2 560 contiguous bytes per lane (8 entries x
5 coordinates x 64 bytes: a gather reads exactly its 320 bytes, every row address is an immediate offset), and a window's entry is gathered BEFORE
the window's four doublings, i.e. ~12 000 instructions ahead of its use.
Exceptional cases: the accumulator is "not started" until the first non-zero digit (that entry is copied, not added); an addition that meets
acc == +-entry cannot happen for a point of G1 (16 x prefix = +-d has no solution with 0 < |d| <= 8) - Z3 == 0 after an addition raises the
lane's flag and the kernel recomputes the lane with the compiled complete formulas (keys outside G1 that a caller did not subgroup-check).

`--selftest`: asmlib's interpreter runs the generated blocks (doubling, addition, table preparation) through the whole algorithm for random
scalars against big-integer group arithmetic; the digit extraction and the loop control are raw text, covered by the GPU parity tests
(tests/test_gpu_batch.py: every [r_i]PK_i sample against the oracle)."""
import argparse
import random
import sys

from asmlib import Asm, Builder, Fp, Machine, MASK, N0, NL, ONE, P, PL, R, RECIP, blk, check_limbs, get, limbs_of, mmul, put

# ---- register plan (VGPRs only) --------------------------------------------------------------------------------------------------------
AX, AY, AZ = blk(0), blk(14), blk(28)
E0 = [42, 56, 70, 84, 98]                                      # the gathered table entry: X, Y, Z, Z^2, Z^3 (14 limbs each: the last row of a block is a 2-word load)
EX, EY, EZ, EZZ, EZZZ = (blk(b) for b in E0)
M_REGS = [112 + i for i in range(NL)]
T = [blk(126 + 14 * i) for i in range(5)]
SX, SY = blk(196), blk(210)                                    # operand slots of the two shared multiplier bodies (SY: also every squaring's doubled operand)
ACC, TMP, TMP2 = 224, 226, 227
V_KLO, V_KHI, V_D, V_S, V_ST, V_FLAG, V_TA, V_COL, V_CARRY, V_OFF, V_LDS, V_T = 228, 229, 230, 231, 232, 233, 234, 236, 238, 239, 240, 241
CLOBBER_V = 242
S_P, S_N0, S_MASK, S_RECIP = 36, 50, 51, 52
S_TAB, S_STR, S_ESTR, S_J, S_K, S_OUT, S_OSTR, S_GP, S_T, S_EXEC, S_M, S_M2, S_SH, S_MD, S_SV, S_STM, S_AD = 54, 56, 58, 60, 61, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80, 82, 84
S_M3, S_M4, S_M5, S_R1 = 86, 88, 90, 92                        # addresses of MUL1, SQR1, PREP; return address of the two leaf bodies
CLOBBER_S = (36, 94)
PIN = ((2, 0), (2, 1), (2, 0))                                 # bounds of the accumulator / of a table entry's X, Y (possibly negated), Z


def new_asm():
    return Asm(ACC, S_P, S_N0, S_MASK, S_RECIP)


def builder(a):
    return Builder(a, M_REGS, TMP, TMP2)


def acc_in():
    return tuple(r.like(*bd) for r, bd in zip((AX, AY, AZ), PIN))


def gen_dbl():
    """acc <- 2 acc, multiplier bodies in place (the hot block: 68 times per scalar)"""
    a = new_asm(); b = builder(a)
    x, y, z = acc_in()
    T0, T1, T2, T3, T4 = T
    A = b.sqr(x, SY, T0)
    Bq = b.sqr(y, SY, T1)
    z3 = b.dot([(y, b.shl(T2, z, 1))], AZ)                      # 2 Y Z as Y (2 Z): canonical
    xb = b.dot([(x, Bq)], T2)
    D = b.carry(T2, b.shl(T2, xb, 2))                           # 4 X B
    Ee = b.carry(T0, b.mul3(T0, A))                             # 3 A
    F = b.sqr(Ee, SY, T3)
    x3 = b.reduce(AX, b.sub_nc(T3, F, b.shl(T4, D, 1)))
    W = b.sub_nc(T2, D, x3)
    nb8 = b.reduce(T4, Bq, -8)
    y3 = b.dot([(Ee, W), (nb8, Bq)], AY)                        # E (D - X3) - 8 B^2
    return a.ins, (x3, y3, z3)


def sub_mul1():
    a = new_asm(); builder(a).dot_body([(SX, SY)], SX)
    return a.ins


def sub_sqr1():
    a = new_asm(); builder(a).sqr_body(SX, SY, SX)
    return a.ins


class Slots:
    """products through the two shared bodies: operands copied into SX / SY where they are not there already, the result left in SX or copied out"""

    def __init__(self):
        self.a = new_asm(); self.b = builder(self.a)

    def mul(self, dst, x, y):
        from asmlib import dot_bounds_ok
        assert dot_bounds_ok([(x, y)])
        self.b.mov(SX, x); self.b.mov(SY, y)
        self.a.e("call", "MUL1", S_R1, S_M3)
        return self.b.mov(dst, SX.like(2, 0))

    def sq(self, dst, x):
        assert x.vb * x.vb <= 2048 and x.lb <= 2
        self.b.mov(SX, x)
        self.a.e("call", "SQR1", S_R1, S_M4)
        return self.b.mov(dst, SX.like(2, 0))


def gen_add():
    """acc <- acc + entry (X2, +-Y2, Z2, Z2^2, Z2^3 in the E registers); V_T <- zero where Z3 == 0.  20 times per scalar: its twelve single products go
    through the two shared bodies (with them in place this block alone is 49 KB and the loop no longer fits the instruction cache: 14 % of the wave cycles
    waited for instructions, profiles/r05_ab)"""
    s_ = Slots(); a, b = s_.a, s_.b
    x1, y1, z1 = acc_in()
    x2, y2, z2 = EX.like(2, 0), EY.like(2, 1), EZ.like(2, 0)
    zz2, zzz2 = EZZ.like(2, 0), EZZZ.like(2, 0)
    T0, T1, T2, T3, T4 = T
    z1z1 = s_.sq(T0, z1)
    u2 = s_.mul(T1, x2, z1z1)
    t = s_.mul(SX, y2, z1)
    s2 = s_.mul(T2, t, z1z1)
    z12 = s_.mul(T0, z1, z2)
    u1 = s_.mul(T3, x1, zz2)
    H = b.sub_nc(T1, u2, u1)
    s1 = s_.mul(T4, y1, zzz2)
    rr = b.sub_nc(T2, s2, s1)
    z3 = s_.mul(AZ, z12, H)
    b.zero_test(V_T, [z3], [T0])
    HH = s_.sq(T0, H)
    HHH = s_.mul(T1, H, HH)
    V = s_.mul(T3, u1, HH)
    r2 = s_.sq(T0, rr)
    tt = b.sub_nc(T0, r2, HHH)
    tt = b.sub_nc(T0, tt, b.shl(SY, V, 1))
    x3 = b.reduce(AX, tt)
    vx = b.sub_nc(T3, V, x3)
    ns1 = b.neg(SY, s1)
    y3 = b.dot([(rr, vx), (ns1, HHH)], AY)
    return a.ins, (x3, y3, z3)


def gen_prep():
    """Z^2 -> T0, Z^3 -> T1 of the accumulator (what a table entry carries besides X, Y, Z)"""
    s_ = Slots()
    x, y, z = acc_in()
    zz = s_.sq(T[0], z)
    s_.mul(T[1], z, zz)
    return s_.a.ins


def gen_first():
    """acc <- the entry (X, +-Y, Z)"""
    a = new_asm(); b = builder(a)
    b.mov(AX, EX.like(2, 0)); b.mov(AY, EY.like(2, 1)); b.mov(AZ, EZ.like(2, 0))
    return a.ins


def text_of_list(ins):
    a = new_asm(); a.ins = ins
    return a.text()


# ---- reference: Jacobian G1 arithmetic on big integers (Montgomery images; the formulas above) ---------------------------------------------
def rdbl(Pt):
    x, y, z = Pt
    A, B_ = mmul(x, x), mmul(y, y)
    D = 4 * mmul(x, B_) % P
    E_ = 3 * A % P
    x3 = (mmul(E_, E_) - 2 * D) % P
    return (x3, (mmul(E_, (D - x3) % P) - 8 * mmul(B_, B_)) % P, 2 * mmul(y, z) % P)


def radd(P1, P2):
    x1, y1, z1 = P1
    x2, y2, z2 = P2
    zz2 = mmul(z2, z2); zzz2 = mmul(z2, zz2)
    z1z1 = mmul(z1, z1)
    u2, s2 = mmul(x2, z1z1), mmul(mmul(y2, z1), z1z1)
    u1, s1 = mmul(x1, zz2), mmul(y1, zzz2)
    H, rr = (u2 - u1) % P, (s2 - s1) % P
    HH = mmul(H, H)
    HHH, V = mmul(H, HH), mmul(u1, HH)
    x3 = (mmul(rr, rr) - HHH - 2 * V) % P
    return (x3, (mmul(rr, (V - x3) % P) - mmul(s1, HHH)) % P, mmul(mmul(z1, z2), H))


def same_point(Pa, Pb):
    za, zb = mmul(Pa[2], Pa[2]), mmul(Pb[2], Pb[2])
    return mmul(Pa[0], zb) == mmul(Pb[0], za) and mmul(Pa[1], mmul(zb, Pb[2])) == mmul(Pb[1], mmul(za, Pa[2]))


def selftest(seed=4):
    rnd = random.Random(seed)
    dbl, dout = gen_dbl()
    add, aout = gen_add()
    prep, first = gen_prep(), gen_first()
    for outs in (dout, aout):
        for o_, bd in zip(outs, PIN):
            assert o_.vb <= bd[0] and o_.lb <= bd[1], (o_.vb, o_.lb, bd)
    mach = Machine(new_asm(), {"MUL1": sub_mul1(), "SQR1": sub_sqr1()})
    R1 = R % P

    def acc_regs():
        return tuple(get(mach, r) % P for r in (AX, AY, AZ))

    def set_entry(ent, neg):
        x, y, z = ent[0]
        for reg, val in zip((EX, EZ, EZZ, EZZZ), (x, z, ent[1], ent[2])):
            put(mach, reg, limbs_of(val))
        ly = limbs_of(y)
        put(mach, EY, [(-l) & 0xffffffff for l in ly] if neg else ly)

    for trial in range(3):
        pk = (rnd.randrange(P), rnd.randrange(P), R1)                   # any triple with Z = 1: the formulas are polynomial identities
        r = rnd.getrandbits(64) | 1
        # ---- the table 1 .. 8 times PK, each entry with Z^2, Z^3, built by the generated blocks
        def run_prep():
            mach.run(prep)
            return (acc_regs(), get(mach, T[0]) % P, get(mach, T[1]) % P)
        for reg, val in zip((AX, AY, AZ), pk):
            put(mach, reg, limbs_of(val))
        table = [run_prep()]
        ref = [pk]
        base = table[0]
        def load_acc(pt):
            for reg, val in zip((AX, AY, AZ), pt):
                put(mach, reg, limbs_of(val))
        for i in range(1, 8):
            if i & 1:                                                    # 2, 4, 6, 8 times PK: the double of entry (i + 1) / 2
                load_acc(table[(i - 1) // 2][0]); mach.run(dbl); ref.append(rdbl(ref[(i - 1) // 2]))
            else:                                                        # 3, 5, 7: the previous entry + PK
                load_acc(table[i - 1][0]); set_entry(base, False); mach.run(add); ref.append(radd(ref[i - 1], pk))
                assert mach.v[V_T] != 0
            assert acc_regs() == ref[i], ("table entry", i + 1)
            for reg, bd in zip((AX, AY, AZ), PIN):
                check_limbs(mach, reg, bd[1])
            table.append(run_prep())
            assert table[i][1] == mmul(ref[i][2], ref[i][2]) and table[i][2] == mmul(ref[i][2], table[i][1])
        # ---- the windows
        kp = r + 0x8888888888888888
        digs = [((kp >> (4 * j)) & 15) - 8 for j in range(16)] + [kp >> 64]
        assert sum(d * 16 ** j for j, d in enumerate(digs)) == r
        started, acc_ref, n_add = False, None, 0
        for j in range(16, -1, -1):
            if j < 16:
                for _ in range(4):
                    mach.run(dbl)
                    if started:
                        acc_ref = rdbl(acc_ref)
            d = digs[j]
            if d == 0:
                continue
            ent = table[abs(d) - 1]
            set_entry(ent, d < 0)
            eref = (ent[0][0], (-ent[0][1]) % P if d < 0 else ent[0][1], ent[0][2])
            if not started:
                mach.run(first); acc_ref = eref; started = True
            else:
                mach.run(add); acc_ref = radd(acc_ref, eref); n_add += 1
                assert mach.v[V_T] != 0, "a point of prime order never meets acc == +-entry"
            assert acc_regs() == acc_ref, ("window", j)
        # as a group element: r * PK by plain double-and-add with the same formulas
        want = None
        for bit in range(63, -1, -1):
            if want is not None:
                want = rdbl(want)
            if (r >> bit) & 1:
                want = pk if want is None else (radd(want, pk) if not same_point(want, pk) else rdbl(want))
        # (random triples are not curve points: the two computations agree as polynomial identities only on a curve; checked on the GPU against the oracle)
    # the flag: adding an entry to itself
    load_acc(table[2][0]); set_entry(table[2], False); mach.run(add)
    assert mach.v[V_T] == 0, "acc == entry must be detected (Z3 == 0)"
    c0 = dict(mach.count); mach.run(dbl); c1 = dict(mach.count); mach.run(add); c2 = dict(mach.count)
    nd, md, na, ma = c1["valu"] - c0["valu"], c1["mad"] - c0["mad"], c2["valu"] - c1["valu"], c2["mad"] - c1["mad"]
    print("gen_pkmul_asm selftest ok: doubling %d instructions (%d multiply-adds), addition %d (%d); per scalar ~%d instructions, %.1f %% multiply-adds"
          % (nd, md, na, ma, 68 * nd + 20 * na + 8 * 900, 100.0 * (68 * md + 20 * ma + 8 * 700) / (68 * nd + 20 * na + 8 * 900)))


# ---- text ---------------------------------------------------------------------------------------------------------------------------
ENTRY_BYTES, LANE_BYTES = 320, 8 * 320


def rows_uniform(store, blocks, e):
    """the rows of table entry e (wave-uniform index): address = this lane's table (V_COL, 64-bit) + an immediate"""
    t = []
    for c, base in enumerate(blocks):
        for q in range(4):
            n = 4 if q < 3 else 2
            r = base + 4 * q
            off = e * ENTRY_BYTES + 64 * c + 16 * q
            if store:
                t.append("global_store_dwordx%d v[%d:%d], v[%d:%d], off offset:%d" % (n, V_COL, V_COL + 1, r, r + n - 1, off))
            else:
                t.append("global_load_dwordx%d v[%d:%d], v[%d:%d], off offset:%d" % (n, r, r + n - 1, V_COL, V_COL + 1, off))
    return t


def store_entry(e):
    return ["s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_T, S_T + 1, S_M5, S_M5 + 1)] + rows_uniform(True, [AX.r[0], AY.r[0], AZ.r[0], T[0].r[0], T[1].r[0]], e)


def load_acc_entry(e):
    return ["s_waitcnt vmcnt(0)"] + rows_uniform(False, [AX.r[0], AY.r[0], AZ.r[0]], e) + ["s_waitcnt vmcnt(0)"]


def kernel_text():
    """operands: %0 flag out (v); %1, %2 the scalar r (v, v: low, high word); %3 table buffer (s pair: 2 560 bytes per lane); %4 unused (s); %5 16 * the lane's
    index (v); %6 output P (s pair: SoA, three planes); %7 its row stride in bytes (s); %8 LDS address of the slot holding the key (x, y) in fp2_lds_put's
    layout (s)"""
    dbl, _ = gen_dbl()
    add, _ = gen_add()
    Tx = []
    Tx += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    Tx += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK), "s_mov_b32 s%d, 0x%x" % (S_RECIP, RECIP)]
    Tx += ["s_mov_b64 s[%d:%d], %%3" % (S_TAB, S_TAB + 1), "s_mov_b32 s%d, %d" % (S_ESTR, ENTRY_BYTES),
           "s_mov_b64 s[%d:%d], %%6" % (S_OUT, S_OUT + 1), "s_mov_b32 s%d, %%7" % S_OSTR, "s_mov_b64 s[%d:%d], exec" % (S_EXEC, S_EXEC + 1)]
    Tx += ["v_mov_b32_e64 v%d, %%5" % V_OFF, "v_mov_b32_e64 v%d, 0" % V_FLAG, "v_mov_b32_e64 v%d, 0" % V_ST]
    # k' = r + 0x8888888888888888, the carry is digit 16
    Tx += ["v_mov_b32_e32 v%d, 0x88888888" % V_T, "v_add_co_u32_e32 v%d, vcc, v%d, %%1" % (V_KLO, V_T), "v_addc_co_u32_e32 v%d, vcc, v%d, %%2, vcc" % (V_KHI, V_T),
           "v_cndmask_b32_e64 v%d, 0, 1, vcc" % V_CARRY]
    # this lane's column as a 64-bit address: table base + 16 * column
    Tx += ["v_lshrrev_b32_e64 v%d, 4, v%d" % (V_T, V_OFF), "s_mov_b32 s%d, %d" % (S_SH, LANE_BYTES),
           "v_mad_u64_u32 v[%d:%d], vcc, v%d, s%d, 0" % (V_COL, V_COL + 1, V_T, S_SH),
           "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (V_COL, V_COL + 1, V_COL, V_COL + 1, S_TAB, S_TAB + 1)]
    # the key from LDS: 28 words (x limbs, y limbs) -> T0, T1 (contiguous), then the accumulator = (x, y, 1) = entry 1; the E registers keep it for the odd entries
    Tx += ["v_mbcnt_lo_u32_b32 v%d, -1, 0" % TMP, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (TMP, TMP), "v_lshlrev_b32_e64 v%d, 4, v%d" % (TMP, TMP),
           "v_add_u32_e64 v%d, %%8, v%d" % (V_LDS, TMP)]
    Tx += ["ds_read_b128 v[%d:%d], v%d offset:%d" % (T[0].r[0] + 4 * q, T[0].r[0] + 4 * q + 3, V_LDS, 1024 * q) for q in range(7)] + ["s_waitcnt lgkmcnt(0)"]
    g = new_asm(); b = builder(g)
    b.mov(AX, T[0].like(2, 0)); b.mov(AY, T[1].like(2, 0))
    for r, l in zip(AZ.r, ONE):
        g.e("movi", r, l)
    Tx += g.text()
    dblt, addt = text_of_list(dbl), text_of_list(add)
    flag_t = ["v_cmp_eq_u32_e64 vcc, 0, v%d" % V_T, "v_cndmask_b32_e64 v%d, v%d, 1, vcc" % (V_FLAG, V_FLAG)]
    # the doubling and the addition are subroutine-free blocks used from several places: two local "calls" through s_setpc keep ONE copy of each
    Tx += ["s_getpc_b64 s[%d:%d]" % (S_M, S_M + 1), ".Lpk_p1%=:", "s_add_u32 s%d, s%d, (.Lpk_dbl%%=-.Lpk_p1%%=)&4294967295" % (S_M, S_M),
           "s_addc_u32 s%d, s%d, (.Lpk_dbl%%=-.Lpk_p1%%=)>>32" % (S_M + 1, S_M + 1),
           "s_getpc_b64 s[%d:%d]" % (S_M2, S_M2 + 1), ".Lpk_p2%=:", "s_add_u32 s%d, s%d, (.Lpk_add%%=-.Lpk_p2%%=)&4294967295" % (S_M2, S_M2),
           "s_addc_u32 s%d, s%d, (.Lpk_add%%=-.Lpk_p2%%=)>>32" % (S_M2 + 1, S_M2 + 1)]
    for nm, sr in (("mul1", S_M3), ("sqr1", S_M4), ("prep", S_M5)):
        Tx += ["s_getpc_b64 s[%d:%d]" % (sr, sr + 1), ".Lpk_q%s%%=:" % nm, "s_add_u32 s%d, s%d, (.Lpk_%s%%=-.Lpk_q%s%%=)&4294967295" % (sr, sr, nm, nm),
               "s_addc_u32 s%d, s%d, (.Lpk_%s%%=-.Lpk_q%s%%=)>>32" % (sr + 1, sr + 1, nm, nm)]
    Tx += ["s_branch .Lpk_main%=",
           ".Lpk_mul1%=:"] + text_of_list(sub_mul1()) + ["s_setpc_b64 s[%d:%d]" % (S_R1, S_R1 + 1),
           ".Lpk_sqr1%=:"] + text_of_list(sub_sqr1()) + ["s_setpc_b64 s[%d:%d]" % (S_R1, S_R1 + 1),
           ".Lpk_dbl%=:"] + dblt + ["s_setpc_b64 s[%d:%d]" % (S_T, S_T + 1), ".Lpk_add%=:"] + addt + flag_t + ["s_setpc_b64 s[%d:%d]" % (S_T, S_T + 1),
           ".Lpk_prep%=:"] + text_of_list(gen_prep()) + ["s_setpc_b64 s[%d:%d]" % (S_T, S_T + 1), ".Lpk_main%=:"]
    Tx += store_entry(0)
    g = new_asm(); b = builder(g)
    b.mov(EX, AX.like(2, 0)); b.mov(EY, AY.like(2, 0)); b.mov(EZ, AZ.like(1, 0)); b.mov(EZZ, T[0].like(2, 0)); b.mov(EZZZ, T[1].like(2, 0))
    Tx += g.text()
    cdbl = ["s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_T, S_T + 1, S_M, S_M + 1)]
    cadd = ["s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_T, S_T + 1, S_M2, S_M2 + 1)]
    # ---- the table: 2 = dbl(1), 3 = 2 + 1, 4 = dbl(2), 5 = 4 + 1, 6 = dbl(3), 7 = 6 + 1, 8 = dbl(4)
    Tx += cdbl + store_entry(1)
    Tx += cadd + store_entry(2)
    Tx += load_acc_entry(1) + cdbl + store_entry(3)
    Tx += cadd + store_entry(4)
    Tx += load_acc_entry(2) + cdbl + store_entry(5)
    Tx += cadd + store_entry(6)
    Tx += load_acc_entry(3) + cdbl + store_entry(7)
    Tx += ["s_waitcnt vmcnt(0)"]
    # ---- the windows, j = 16 .. 0
    Tx += ["s_mov_b32 s%d, 16" % S_J, ".Lpk_win%=:"]
    # digit: nibble j of k' (j = 16: the carry + 8), d = nibble - 8, sign mask, magnitude
    Tx += ["s_cmp_eq_u32 s%d, 16" % S_J, "s_cbranch_scc1 .Lpk_top%=",
           "s_lshl_b32 s%d, s%d, 2" % (S_SH, S_J), "v_lshrrev_b64 v[%d:%d], s%d, v[%d:%d]" % (V_TA, V_TA + 1, S_SH, V_KLO, V_KHI),
           "v_and_b32_e64 v%d, 15, v%d" % (V_D, V_TA), "s_branch .Lpk_dig%=",
           ".Lpk_top%=:", "v_add_u32_e64 v%d, 8, v%d" % (V_D, V_CARRY), ".Lpk_dig%=:",
           "v_subrev_u32_e64 v%d, 8, v%d" % (V_D, V_D), "v_ashrrev_i32_e64 v%d, 31, v%d" % (V_S, V_D),
           "v_xor_b32_e64 v%d, v%d, v%d" % (V_D, V_D, V_S), "v_sub_u32_e64 v%d, v%d, v%d" % (V_D, V_D, V_S)]
    # gather of the entry |d| (lanes with d != 0) BEFORE the doublings: address = column + (|d| - 1) * 20 rows
    Tx += ["v_cmp_ne_u32_e64 s[%d:%d], 0, v%d" % (S_MD, S_MD + 1, V_D), "s_and_saveexec_b64 s[%d:%d], s[%d:%d]" % (S_SV, S_SV + 1, S_MD, S_MD + 1),
           "s_cbranch_execz .Lpk_nog%=",
           "v_add_u32_e64 v%d, -1, v%d" % (V_T, V_D),
           "v_mad_u64_u32 v[%d:%d], vcc, v%d, s%d, v[%d:%d]" % (V_TA, V_TA + 1, V_T, S_ESTR, V_COL, V_COL + 1)]
    for c, base in enumerate(E0):
        for q in range(4):
            n = 4 if q < 3 else 2
            Tx.append("global_load_dwordx%d v[%d:%d], v[%d:%d], off offset:%d" % (n, base + 4 * q, base + 4 * q + n - 1, V_TA, V_TA + 1, 64 * c + 16 * q))
    Tx += [".Lpk_nog%=:", "s_mov_b64 exec, s[%d:%d]" % (S_SV, S_SV + 1)]
    # four doublings (none in front of digit 16)
    Tx += ["s_cmp_eq_u32 s%d, 16" % S_J, "s_cbranch_scc1 .Lpk_nodbl%=", "s_mov_b32 s%d, 4" % S_K, ".Lpk_d4%=:"] + cdbl + \
          ["s_sub_u32 s%d, s%d, 1" % (S_K, S_K), "s_cmp_lg_u32 s%d, 0" % S_K, "s_cbranch_scc1 .Lpk_d4%=", ".Lpk_nodbl%=:", "s_waitcnt vmcnt(0)"]
    # the entry's sign: Y <- (Y ^ s) - s
    for i in range(NL):
        Tx += ["v_xor_b32_e64 v%d, v%d, v%d" % (EY.r[i], EY.r[i], V_S), "v_sub_u32_e64 v%d, v%d, v%d" % (EY.r[i], EY.r[i], V_S)]
    # started lanes add, the others take the entry as their first value
    Tx += ["v_cmp_ne_u32_e64 s[%d:%d], 0, v%d" % (S_STM, S_STM + 1, V_ST),
           "s_and_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (S_AD, S_AD + 1, S_MD, S_MD + 1, S_STM, S_STM + 1),
           "s_andn2_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (S_MD, S_MD + 1, S_MD, S_MD + 1, S_STM, S_STM + 1),
           "s_mov_b64 s[%d:%d], exec" % (S_SV, S_SV + 1),
           "s_and_b64 exec, s[%d:%d], s[%d:%d]" % (S_SV, S_SV + 1, S_AD, S_AD + 1), "s_cbranch_execz .Lpk_noadd%="] + cadd + [".Lpk_noadd%=:",
           "s_and_b64 exec, s[%d:%d], s[%d:%d]" % (S_SV, S_SV + 1, S_MD, S_MD + 1), "s_cbranch_execz .Lpk_nofirst%="] + text_of_list(gen_first()) + \
          ["v_mov_b32_e64 v%d, 1" % V_ST, ".Lpk_nofirst%=:", "s_mov_b64 exec, s[%d:%d]" % (S_SV, S_SV + 1),
           "s_sub_u32 s%d, s%d, 1" % (S_J, S_J), "s_cmp_ge_i32 s%d, 0" % S_J, "s_cbranch_scc1 .Lpk_win%="]
    # a lane that never started (r = 0) holds the point at infinity
    Tx += ["v_cmp_eq_u32_e64 vcc, 0, v%d" % V_ST, "s_and_saveexec_b64 s[%d:%d], vcc" % (S_SV, S_SV + 1)]
    Tx += ["v_mov_b32_e64 v%d, 0" % r for r in AX.r + AY.r + AZ.r]
    Tx += ["s_mov_b64 exec, s[%d:%d]" % (S_SV, S_SV + 1)]
    # the result, SoA: three planes of four rows
    Tx += ["s_mov_b64 s[%d:%d], s[%d:%d]" % (S_GP, S_GP + 1, S_OUT, S_OUT + 1)]
    for src in (AX, AY, AZ):
        for q in range(4):
            n = 4 if q < 3 else 2
            r = src.r[4 * q]
            Tx.append("global_store_dwordx%d v%d, v[%d:%d], s[%d:%d]" % (n, V_OFF, r, r + n - 1, S_GP, S_GP + 1))
            Tx += ["s_add_u32 s%d, s%d, s%d" % (S_GP, S_GP, S_OSTR), "s_addc_u32 s%d, s%d, 0" % (S_GP + 1, S_GP + 1)]
    Tx += ["s_waitcnt vmcnt(0)", "s_mov_b64 exec, s[%d:%d]" % (S_EXEC, S_EXEC + 1), "v_mov_b32_e64 %%0, v%d" % V_FLAG]
    return Tx


def clobbers():
    c = ["v%d" % i for i in range(CLOBBER_V)] + ["s%d" % i for i in range(*CLOBBER_S)] + ["vcc", "scc", "memory"]
    return ", ".join('"%s"' % x for x in c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    if a.selftest:
        selftest()
        return
    lines = [l if l.startswith(".L") else "\\t" + l for l in kernel_text()]
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_pkmul_asm.py -- do not edit.\n"
           "// operands: %0 flag out (v), %1 %2 the scalar (v, v), %3 table scratch (s pair), %4 its row stride in bytes (s), %5 16 * column (v), %6 output P (s pair), %7 its row stride in bytes (s), %8 LDS address of the key's slot (s)\n"
           "#define BLS_PKMUL_ASM_BODY \\\n" + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_PKMUL_ASM_CLOBBERS " + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
