#!/usr/bin/env python3
"""Generates k_hash_clear's body (csrc/kernels.hip) as ONE hand-allocated gfx950 inline-asm statement: per lane, the two mapped points
q0, q1 of a message (k_hash_map's output) -> H = clear_cofactor(q0 + q1), the last stage of hash-to-G2 (h2c.hpp clear_cofactor_g2_chain,
RFC 9380 G.3 / Budroni-Pintore):  P = q0 + q1;  c = [|x|] P;  u = psi^2(2P) - psi(P) + c - P;  base = psi(P) - c;  H = u - [|x|] base
- two chains of 63 doublings + 5 additions, 7 more additions, one doubling, three psi maps: everything from the loads to the store.

Round 5 first moved only the chains into assembly and gained nothing: the compiled chain was already at ~4.2 cycles per instruction; what
ran at 7.3 was the compiled REST - seven out-of-line complete additions whose 252 argument words travel through scratch memory, 11 % of
the instructions and 17 % of the kernel's time (profiles/r05_ab).  Hence the whole kernel:
  doubling   curve.hpp jac_dbl_lazy:  A = X^2, B = Y^2, D = 4 X B, E = 3 A, X3 = E^2 - 2 D, Z3 = Y (2 Z) (canonical, no carry),
             Y3 = E (D - X3) - 8 B^2 as two three-term lazily reduced dot products; the scaled terms -8 (B0 + B1), -16 B0 come out of
             fp_reduce's own multiply-add chain (asmlib reduce with a folded scale);
  addition   curve.hpp jac_precompute + jac_add_pre for EVERY addition: the second operand's Z^2, Z^3 once (PREP), then acc + base with
             3 squarings + 11 products (ADD); a chain adds the same base five times, the other seven additions pay PREP each - the
             4 + 12 of a plain Jacobian addition.  The exceptional cases of a complete addition (an operand at infinity, P = +-Q) all end
             in Z3 = 0: PREP tests the base's Z, ADD tests Z3, either raises a per-lane FLAG and the kernel recomputes a flagged lane with
             the compiled complete formulas (never seen in practice: hash outputs);
  psi        (conj(x) cx, conj(y) cy, conj(z)): two products by constants.
Register plan as gen_lines_asm.py: operand slots SA, SB, result block R0, K; leaf subroutines SQR, MUL, DOT3A / DOT3B (three-term dot
products: SA0 SB0 + SA1 SB1 + R0 K and the swapped pairing, so that W = D - X3 serves both halves of Y3 from one place); second-level
subroutines PREP, ADD, PSI, third-level CHAIN.  The accumulator lives in VGPR blocks X, Y, Z ("homes"), the base with its Z^2, Z^3 and
the addition's parked intermediates in AGPRs, a second operand is staged in (B, T, SA).  Points that must outlive a chain wait in three
LDS slots (P, later u) and in two per-lane columns of a global scratch area (the context's line store, unused at this stage).

`--selftest`: asmlib's interpreter against big-integer arithmetic - single steps with their bounds, the chain, and the WHOLE kernel body
(memory operations emulated) checked as a group element against the reference formula on random points of E2'(Fp2)-like triples, plus
the flag on crafted exceptional inputs (tests/test_asm_loops.py).
"""
import argparse
import random
import sys

from asmlib import (Asm, Builder, F2, Fp, Fp2, Machine, N0, NL, P, PL, R, RECIP, X_ABS, MASK, blk, blk2, check_limbs, dot_bounds_ok, get, limbs_of,
                    mmul, put)
import gen_lines_asm as gl
from gen_lines_asm import SA, SB, R0, K, M_REGS, FREE, ACC, TMP, TMP2, S_P, S_N0, S_MASK, S_RECIP, new_asm, builder

# ---- register plan (VGPR slots and scratch as gen_lines_asm) --------------------------------------------------------------------------
X, Y, Z, B, T = blk2(FREE[0]), blk2(FREE[2]), blk2(FREE[4]), blk2(FREE[6]), blk2(FREE[8])
HOMES, STAGE = (X, Y, Z), (B, T, SA)
V_LDS, V_FLAG, V_OFFM, V_OFFH = 242, 243, 244, 245
CLOBBER_V = 250
ABX, ABY, ABZ = blk2(0, True), blk2(28, True), blk2(56, True)              # the base point (second operand of ADD)
AZZ, AZZZ = blk2(84, True), blk2(112, True)                                 # its Z^2, Z^3
AP = [blk2(140 + 28 * i, True) for i in range(4)]                           # parked intermediates of the addition (a140 .. a251)
S_SQR, S_MUL, S_D3A, S_D3B, S_PREP, S_ADD, S_PSI, S_CHAIN = 54, 56, 58, 60, 62, 64, 66, 68
S_RET, S_RET2, S_RET3 = 70, 72, 74
S_I, S_XA, S_T, S_GP, S_MSTR, S_HSTR, S_SSTR = 76, 78, 80, 82, 84, 85, 86
S_MB, S_HB, S_SB = 88, 90, 92                    # bases (pairs): mapped points, output, scratch columns
CLOBBER_S = (36, 96)

# psi constants (Montgomery images): cx = 1 / xi^((p-1)/3), cy = 1 / xi^((p-1)/2), xi = 1 + u   (tools/gen_constants.py PSI_CX, PSI_CY)
def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = _f2mul(r, a)
        a = _f2mul(a, a)
        e >>= 1
    return r


def _f2inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, -a[1] * n % P)


PSI_CX = tuple(c * R % P for c in _f2inv(_f2pow((1, 1), (P - 1) // 3)))
PSI_CY = tuple(c * R % P for c in _f2inv(_f2pow((1, 1), (P - 1) // 2)))
# input bounds of the homes / of a staged second operand: whatever k_hash_map stores (x carried after an addition: |v| < 4 p) and whatever
# a doubling, an addition or psi leaves
PIN = ((4, 1), (4, 1), (2, 1))


def pin(triple):
    return tuple(Fp2(t.c0.like(*bd), t.c1.like(*bd)) for t, bd in zip(triple, PIN))


def sub_dot3(swap):
    """SA0 SB0 + SA1 SB1 + R0 K -> SA0   (swap: SA0 SB1 + SA1 SB0 + R0 K)"""
    def f():
        a = new_asm(); b = builder(a)
        y0, y1 = (SB.c1, SB.c0) if swap else (SB.c0, SB.c1)
        b.dot_body([(SA.c0, y0), (SA.c1, y1), (R0, K)], SA.c0)
        return a.ins
    return f


class Steps(gl.Steps):
    def call(self, name, ret=S_RET):
        self.a.e("call", name, ret, SUB_ADDR[name])

    def DOT3(self, swap, x0, y0, x1, y1, x2, y2):
        assert dot_bounds_ok([(x0, y0), (x1, y1), (x2, y2)]), "DOT3 operand bounds"
        self.call("DOT3B" if swap else "DOT3A")
        return SA.c0.like(2, 0)

    # -- doubling: homes <- 2 homes (curve.hpp jac_dbl_lazy)
    def dbl(self, Pt):
        b = self.b
        Xv, Yv, Zv = Pt
        self.mov2(SA, Yv)
        Bv = self.mov2(B, self.SQR(Yv))
        z2 = Fp2(b.shl(SB.c0, Zv.c0, 1), b.shl(SB.c1, Zv.c1, 1))
        Z3 = self.mov2(Z, self.MUL(Yv, z2))                                     # 2 Y Z as Y * (2 Z): canonical, no carry
        self.mov2(SA, Xv)
        Av = self.SQR(Xv)
        Ev = Fp2(b.carry(T.c0, b.mul3(T.c0, Av.c0)), b.carry(T.c1, b.mul3(T.c1, Av.c1)))
        Bb = self.mov2(SB, Bv)
        XB = self.MUL(Xv, Bb)                                                   # SA still holds X (its second half is the product's, but X is dead)
        Dv = Fp2(b.carry(Y.c0, b.shl(Y.c0, XB.c0, 2)), b.carry(Y.c1, b.shl(Y.c1, XB.c1, 2)))      # D takes Y's block (Y is dead)
        Ea = self.mov2(SA, Ev)
        Fq = self.SQR(Ea)
        X3 = Fp2(b.reduce(X.c0, b.sub_nc(X.c0, Fq.c0, b.shl(SB.c0, Dv.c0, 1))), b.reduce(X.c1, b.sub_nc(X.c1, Fq.c1, b.shl(SB.c1, Dv.c1, 1))))
        W = self.sub2(SB, Dv, X3)
        # Y3.re = E0 W0 - E1 W1 - [8 (B0 + B1)] (B0 - B1)
        ne1 = b.neg(SA.c1, Ea.c1)
        x2 = b.reduce(R0, b.add_nc(R0, Bv.c0, Bv.c1), -8)
        y2 = b.sub_nc(K, Bv.c0, Bv.c1)
        Y3re = b.mov(Y.c0, self.DOT3(False, Ea.c0, W.c0, ne1, W.c1, x2, y2))
        # Y3.im = E0 W1 + E1 W0 - [16 B0] B1
        e0 = b.mov(SA.c0, Ev.c0)
        e1 = b.neg(SA.c1, ne1)
        x2 = b.reduce(R0, Bv.c0, -16)
        y2 = b.mov(K, Bv.c1)
        Y3im = b.mov(Y.c1, self.DOT3(True, e0, W.c1, e1, W.c0, x2, y2))
        return (X3, Fp2(Y3re, Y3im), Z3)

    # -- PREP: the staged point (B, T, SA) becomes the base: copies, Z^2, Z^3, flag if it is the point at infinity
    def prep(self):
        b = self.b
        Xs, Ys, Zs = pin(STAGE)
        for src, dst in ((Xs, ABX), (Ys, ABY), (Zs, ABZ)):
            self.mov2(dst, src)
        b.flag_if_zero(V_FLAG, [Zs.c0, Zs.c1], [R0, K])
        zz = self.SQR(Zs)                                      # SA holds Z already
        zzb = self.mov2(SB, zz)
        self.mov2(AZZ, zz)
        zzz = self.MUL(Zs, zzb)
        self.mov2(AZZZ, zzz)

    # -- ADD: homes <- homes + base (curve.hpp jac_add_pre); every exceptional case ends in Z3 = 0 -> flag
    def add(self, Pt):
        b = self.b
        X1, Y1, Z1 = Pt
        bx, by, bz = pin((ABX, ABY, ABZ))
        zz2, zzz2 = Fp2(AZZ.c0.like(2, 0), AZZ.c1.like(2, 0)), Fp2(AZZZ.c0.like(2, 0), AZZZ.c1.like(2, 0))
        P0, P1, P2, P3 = AP
        self.mov2(SA, Z1)
        z1z1 = self.SQR(Z1)
        self.mov2(P0, z1z1)
        zb = self.mov2(SB, z1z1)
        self.mov2(SA, bx)
        U2 = self.mov2(B, self.MUL(bx, zb))
        self.mov2(SA, by)
        z1b = self.mov2(SB, Z1)
        t = self.MUL(by, z1b)
        ta = self.mov2(SA, t)
        zb = self.mov2(SB, Fp2(P0.c0.like(2, 0), P0.c1.like(2, 0)))
        S2 = self.mov2(T, self.MUL(ta, zb))
        # Z1 Z2 (Z1 is read for the last time), then U1, H
        self.mov2(SA, Z1)
        bzb = self.mov2(SB, bz)
        z1z2 = self.mov2(P0, self.MUL(Z1, bzb))
        self.mov2(SA, X1)
        zzb = self.mov2(SB, zz2)
        U1 = self.MUL(X1, zzb)
        self.mov2(P1, U1)
        Hv = self.carry2(X, self.sub2(X, U2, U1))
        self.mov2(SA, Y1)
        zzzb = self.mov2(SB, zzz2)
        S1 = self.MUL(Y1, zzzb)
        self.mov2(P2, S1)
        rr = self.carry2(Y, self.sub2(Y, S2, S1))
        # Z3 = (Z1 Z2) H, flag
        hb = self.mov2(SB, Hv)
        za = self.mov2(SA, Fp2(P0.c0.like(2, 0), P0.c1.like(2, 0)))
        Z3 = self.mov2(Z, self.MUL(za, hb))
        b.flag_if_zero(V_FLAG, [Z3.c0, Z3.c1], [R0, K])
        # HH, HHH, V
        ha = self.mov2(SA, Hv)
        HH = self.SQR(ha)
        hhb = self.mov2(SB, HH)
        HHH = self.mov2(B, self.MUL(ha, hhb))
        u1a = self.mov2(SA, Fp2(P1.c0.like(2, 0), P1.c1.like(2, 0)))
        V = self.mov2(T, self.MUL(u1a, hhb))
        # X3 = r^2 - HHH - 2 V
        ra = self.mov2(SA, rr)
        r2 = self.SQR(ra)
        t1 = self.sub2(X, r2, HHH)
        v2 = Fp2(b.shl(SB.c0, V.c0, 1), b.shl(SB.c1, V.c1, 1))
        t2 = self.sub2(X, t1, v2)
        X3 = Fp2(b.reduce(X.c0, t2.c0), b.reduce(X.c1, t2.c1))
        # Y3 = r (V - X3) - S1 HHH
        vx = self.sub2(SA, V, X3)
        rb = self.mov2(SB, rr)
        m1 = self.mov2(P3, self.MUL(vx, rb))
        s1a = self.mov2(SA, Fp2(P2.c0.like(2, 0), P2.c1.like(2, 0)))
        hb = self.mov2(SB, HHH)
        m2 = self.MUL(s1a, hb)
        m1b = self.mov2(SB, Fp2(P3.c0.like(2, 0), P3.c1.like(2, 0)))
        Y3 = self.carry2(Y, self.sub2(Y, m1b, m2))
        return (X3, Y3, Z3)

    # -- PSI: homes <- psi(homes) = (conj(x) cx, conj(y) cy, conj(z))
    def psi(self, Pt):
        b = self.b
        out = []
        for src, home, cst in ((Pt[0], X, PSI_CX), (Pt[1], Y, PSI_CY)):
            b.mov(SA.c0, src.c0)
            cj = Fp2(SA.c0.like(src.c0.vb, src.c0.lb), b.neg(SA.c1, src.c1))
            for part, val in ((SB.c0, cst[0]), (SB.c1, cst[1])):
                for r, l in zip(part.r, limbs_of(val)):
                    self.a.e("movi", r, l)
            kc = Fp2(SB.c0.like(1, 0), SB.c1.like(1, 0))
            out.append(self.mov2(home, self.MUL(cj, kc)))
        out.append(Fp2(Z.c0.like(Pt[2].c0.vb, Pt[2].c0.lb), b.neg(Z.c1, Pt[2].c1)))
        return tuple(out)

    # -- small moves between the homes and the staging triple
    def stage_homes(self, neg_y=False):
        for src, dst, ng in ((X, B, False), (Y, T, neg_y), (Z, SA, False)):
            self.b.mov(dst.c0, src.c0.like(1, 1))
            self.b.mov(dst.c1, src.c1.like(1, 1))
            if ng:
                self.b.neg(dst.c0, dst.c0.like(1, 1)); self.b.neg(dst.c1, dst.c1.like(1, 1))

    def neg_y(self, which):
        self.b.neg(which.c0, which.c0.like(1, 1)); self.b.neg(which.c1, which.c1.like(1, 1))


SUB_ADDR = {"SQR": S_SQR, "MUL": S_MUL, "DOT3A": S_D3A, "DOT3B": S_D3B, "PREP": S_PREP, "ADD": S_ADD, "PSI": S_PSI, "CHAIN": S_CHAIN}
LEAVES = {"SQR": gl.sub_sqr, "MUL": gl.sub_mul, "DOT3A": sub_dot3(False), "DOT3B": sub_dot3(True)}


def check_fixed_point(outs):
    for out_ in outs:
        for p_, bd in zip(out_, PIN):
            for c_ in (p_.c0, p_.c1):
                assert c_.vb <= bd[0] and c_.lb <= bd[1], (c_.vb, c_.lb, bd)


def build_level2():
    """instruction lists of PREP, ADD, PSI and of one doubling (bounds: the widest input, PIN; every output stays inside it)"""
    st = Steps()
    st.prep(); prep = st.a.ins; st.a.ins = []
    D = st.dbl(pin(HOMES)); dbl = st.a.ins; st.a.ins = []
    A = st.add(pin(HOMES)); add = st.a.ins; st.a.ins = []
    S = st.psi(pin(HOMES)); psi = st.a.ins; st.a.ins = []
    check_fixed_point((D, A, S))
    return prep, dbl, add, psi


# ---- memory: points in LDS slots and in per-lane columns of global arrays (raw text + an interpreter hook that moves the same data) ---------
def pt_regs(triple):
    return [r for f in triple for c in (f.c0, f.c1) for r in c.r]


def mem_op(a, triple, key, store, text):
    for l in text:
        a.e("raw", l)
    regs = pt_regs(triple)
    if store:
        a.e("hook", lambda m, key=key, regs=regs: m.mem.__setitem__(key, [m.v[r] for r in regs]))
    else:
        def ld(m, key=key, regs=regs):
            for r, val in zip(regs, m.mem[key]):
                m.v[r] = val
        a.e("hook", ld)


def lds_point(a, triple, store):
    if TWO_WAVE:
        return global_point(a, triple, ("lds",), store, S_SB3, S_SSTR, V_OFFH)
    t = []
    for s_, f in enumerate(triple):
        base = f.c0.r[0]
        assert f.c1.r[0] == base + NL
        for q in range(7):
            if store:
                t.append("ds_write_b128 v%d, v[%d:%d] offset:%d" % (V_LDS, base + 4 * q, base + 4 * q + 3, 7168 * s_ + 1024 * q))
            else:
                t.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (base + 4 * q, base + 4 * q + 3, V_LDS, 7168 * s_ + 1024 * q))
    t.append("s_waitcnt lgkmcnt(0)")
    mem_op(a, triple, ("lds",), store, t)


def global_point(a, triple, key, store, s_base, s_stride, v_off, imm=0):
    """24 rows (6 Fp planes x 4 limb groups) of one lane's column: row pointer = base + row * stride, address = row pointer + v_off + imm"""
    t = ["s_mov_b64 s[%d:%d], s[%d:%d]" % (S_GP, S_GP + 1, s_base, s_base + 1)]
    for f in triple:
        for part in (f.c0, f.c1):
            for q in range(4):
                r = part.r[4 * q]
                n = 4 if q < 3 else 2
                if store:
                    t.append("global_store_dwordx%d v%d, v[%d:%d], s[%d:%d] offset:%d" % (n, v_off, r, r + n - 1, S_GP, S_GP + 1, imm))
                else:
                    t.append("global_load_dwordx%d v[%d:%d], v%d, s[%d:%d] offset:%d" % (n, r, r + n - 1, v_off, S_GP, S_GP + 1, imm))
                t += ["s_add_u32 s%d, s%d, s%d" % (S_GP, S_GP, s_stride), "s_addc_u32 s%d, s%d, 0" % (S_GP + 1, S_GP + 1)]
    t.append("s_waitcnt vmcnt(0)")          # loads: before the first use; stores: before the column is read back (rare: a few times per lane)
    mem_op(a, triple, key, store, t)


def main_program(dbl_ins):
    """the kernel body behind the subroutines, as instruction tuples (calls, raw memory text, hooks)"""
    st = Steps()
    a = st.a
    c2 = lambda name: st.call(name, S_RET2 if name in ("PREP", "ADD", "PSI") else S_RET3)
    a.e("movi", V_FLAG, 0)
    global_point(a, HOMES, ("M", 0), False, S_MB, S_MSTR, V_OFFM, 0)               # q0
    global_point(a, STAGE, ("M", 1), False, S_MB, S_MSTR, V_OFFM, 16)              # q1
    c2("PREP"); c2("ADD")                                                           # P = q0 + q1
    lds_point(a, HOMES, True)                                                       # P waits in LDS
    c2("CHAIN")                                                                     # c = [|x|] P
    global_point(a, HOMES, ("S", 0), True, S_SB, S_SSTR, V_OFFH)                    # c -> scratch column 0
    lds_point(a, HOMES, False)
    c2("PSI")                                                                       # t2 = psi(P)
    global_point(a, HOMES, ("S", 1), True, S_SB + 2, S_SSTR, V_OFFH)                # t2 -> scratch column 1 (base pair S_SB+2 = S_SB + 24 rows)
    st.stage_homes(neg_y=True)
    c2("PREP")                                                                      # base = -psi(P)
    lds_point(a, HOMES, False)                                                      # P
    a.ins += dbl_ins                                                                # 2 P (the one doubling outside the chains: a second copy of the body)
    c2("PSI"); c2("PSI")                                                            # psi^2(2 P)
    c2("ADD")                                                                       # u = psi^2(2P) - psi(P)
    global_point(a, STAGE, ("S", 0), False, S_SB, S_SSTR, V_OFFH)
    c2("PREP"); c2("ADD")                                                           # u += c        (- [x] P, x < 0)
    lds_point(a, STAGE, False)
    st.neg_y(T)
    c2("PREP"); c2("ADD")                                                           # u -= P
    lds_point(a, HOMES, True)                                                       # u waits in LDS
    global_point(a, HOMES, ("S", 0), False, S_SB, S_SSTR, V_OFFH)
    st.neg_y(Y)                                                                     # -c = [x] P
    global_point(a, STAGE, ("S", 1), False, S_SB + 2, S_SSTR, V_OFFH)
    c2("PREP"); c2("ADD")                                                           # base = [x] P + psi(P)
    c2("CHAIN")                                                                     # c2 = [|x|] base
    st.neg_y(Y)                                                                     # [x] base
    lds_point(a, STAGE, False)
    c2("PREP"); c2("ADD")                                                           # H = u + [x] base
    global_point(a, HOMES, ("H",), True, S_HB, S_HSTR, V_OFFH)
    return a.ins


def chain_ins(dbl_ins_unused=None):
    """CHAIN: homes <- [|x|] homes.  The loop control is raw text; the interpreter runs the equivalent Python loop (chain_hook)."""
    st = Steps()
    st.stage_homes()
    st.call("PREP", S_RET2)
    return st.a.ins


# ---- reference model: Jacobian group law on y^2 = x^3 + b over Fp2 with Montgomery images (products a b / R) ------------------------------
def ref_dbl(Pt):
    Xr, Yr, Zr = Pt
    A, Bq = Xr.sqr(), Yr.sqr()
    D = (Xr * Bq) * 4
    E = A * 3
    x3 = E.sqr() - D * 2
    y3 = E * (D - x3) - Bq.sqr() * 8
    return (x3, y3, (Yr * Zr) * 2)


def ref_add(P1, P2):
    X1, Y1, Z1 = P1
    X2, Y2, Z2 = P2
    z1z1, z2z2 = Z1.sqr(), Z2.sqr()
    U1, U2 = X1 * z2z2, X2 * z1z1
    S1, S2 = Y1 * (Z2 * z2z2), (Y2 * Z1) * z1z1
    H, rr = U2 - U1, S2 - S1
    HH = H.sqr()
    HHH, V = H * HH, U1 * HH
    x3 = rr.sqr() - HHH - V * 2
    return (x3, rr * (V - x3) - S1 * HHH, (Z1 * Z2) * H)


def ref_neg(Pt):
    return (Pt[0], -Pt[1], Pt[2])


def ref_psi(Pt):
    cj = lambda v: F2(v.c0, -v.c1)
    return (cj(Pt[0]) * F2(*PSI_CX), cj(Pt[1]) * F2(*PSI_CY), cj(Pt[2]))


def ref_chain(base):
    acc = base
    for bit in range(62, -1, -1):
        acc = ref_dbl(acc)
        if (X_ABS >> bit) & 1:
            acc = ref_add(acc, base)
    return acc


def ref_clear(q0, q1):
    Pp = ref_add(q0, q1)
    c = ref_chain(Pp)
    t2 = ref_psi(Pp)
    u = ref_add(ref_psi(ref_psi(ref_dbl(Pp))), ref_neg(t2))
    u = ref_add(u, c)
    u = ref_add(u, ref_neg(Pp))
    base = ref_add(ref_neg(c), t2)
    return ref_add(ref_neg(ref_chain(base)), u)


class M2(Machine):
    def __init__(self, proto, subs):
        Machine.__init__(self, proto, subs)
        self.mem = {}


def selftest(walk=True, seed=5):
    rnd = random.Random(seed)
    prep, dbl, add, psi = build_level2()
    subs = {name: fn() for name, fn in LEAVES.items()}
    subs.update({"PREP": prep, "ADD": add, "PSI": psi})
    mach = M2(new_asm(), subs)

    def chain_hook_ins():
        ins = list(chain_ins())
        for bit in range(62, -1, -1):
            ins += dbl
            if (X_ABS >> bit) & 1:
                ins.append(("call", "ADD", S_RET2, S_ADD))
        return ins
    subs["CHAIN"] = chain_hook_ins()

    def rand_f2():
        return F2(rnd.randrange(P), rnd.randrange(P))

    def load(regs, vals):
        for reg, val in zip(regs, vals):
            put(mach, reg.c0, limbs_of(val.c0)); put(mach, reg.c1, limbs_of(val.c1))

    def check(Tregs, Tref, what):
        for reg, ref, bd in zip(Tregs, Tref, PIN):
            got = F2(get(mach, reg.c0), get(mach, reg.c1))
            assert got == ref, (what, "coordinate mismatch")
            for c in (reg.c0, reg.c1):
                check_limbs(mach, c, bd[1])
                assert abs(get(mach, c)) <= bd[0] * P, (what, "value bound")

    def to_mem(val_triple):
        out = []
        for f in val_triple:
            out += limbs_of(f.c0) + limbs_of(f.c1)
        return out

    # ---- single steps
    base = (rand_f2(), rand_f2(), rand_f2())          # any triples: the formulas are polynomial identities
    acc = (rand_f2(), rand_f2(), rand_f2())
    load(STAGE, base)
    mach.run(prep)
    assert mach.v[V_FLAG] == 0
    load(HOMES, acc)
    n0 = dict(mach.count)
    mach.run(dbl)
    acc = ref_dbl(acc)
    check(HOMES, acc, "dbl")
    n1 = dict(mach.count)
    mach.run(add)
    acc = ref_add(acc, base)
    check(HOMES, acc, "add")
    n2 = dict(mach.count)
    mach.run(psi)
    acc = ref_psi(acc)
    check(HOMES, acc, "psi")
    assert mach.v[V_FLAG] == 0
    # exceptional inputs raise the flag: acc == base (H = 0), base at infinity
    load(HOMES, base)
    mach.run(add)
    assert mach.v[V_FLAG] == 1, "P == Q must raise the flag"
    mach.v[V_FLAG] = 0
    load(STAGE, (base[0], base[1], F2(0, 0)))
    mach.run(prep)
    assert mach.v[V_FLAG] == 1, "a base at infinity must raise the flag"
    mach.v[V_FLAG] = 0
    if walk:
        # ---- the whole kernel body on two random triples
        q0, q1 = (rand_f2(), rand_f2(), rand_f2()), (rand_f2(), rand_f2(), rand_f2())
        mach.mem[("M", 0)], mach.mem[("M", 1)] = to_mem(q0), to_mem(q1)
        c0 = dict(mach.count)
        mach.run(main_program(dbl))
        want = ref_clear(q0, q1)
        got = mach.mem[("H",)]
        from asmlib import value_of
        for k, ref in enumerate(want):
            g = F2(value_of(got[28 * k:28 * k + 14]), value_of(got[28 * k + 14:28 * k + 28]))
            assert g == ref, ("whole kernel", k)
        assert mach.v[V_FLAG] == 0
        tot = {k: mach.count[k] - c0[k] for k in c0}
        print("whole kernel body: %d VALU instructions, %d multiply-adds (%.1f %%)" % (tot["valu"], tot["mad"], 100.0 * tot["mad"] / tot["valu"]))
    d = {k: n1[k] - n0[k] for k in n0}
    a_ = {k: n2[k] - n1[k] for k in n0}
    print("gen_clear_asm selftest ok: doubling %d VALU instructions (%d multiply-adds, %.1f %%, %d calls), addition %d (%d, %d calls); subroutines %s"
          % (d["valu"], d["mad"], 100.0 * d["mad"] / d["valu"], d["calls"], a_["valu"], a_["mad"], a_["calls"],
             {k: len(v) for k, v in subs.items() if k != "CHAIN"}))


# ---- text ---------------------------------------------------------------------------------------------------------------------------
# TWO_WAVE (--two-wave, experiment of round 6): the same instruction lists for a kernel of 256 registers, two waves per SIMD.  Everything the
# one-wave form keeps in AGPRs (the base point with its Z^2, Z^3, the addition's four parked intermediates: 252 words) and in LDS (the point that
# outlives a chain) goes to global memory instead: "AGPR k" becomes row k of a wave-private block (256 contiguous bytes per row: one dword per
# lane), the LDS slots become a third per-lane column of the scratch area.  The doubling loop - 126 of the kernel's 144 steps - touches none of
# it.  The translation is textual (every accvgpr move becomes one dword load / store, a wait in front of and behind every run of loads); the
# interpreter keeps running the AGPR form, which moves the same data.
TWO_WAVE = False
V_WOFF = 246                                   # this lane's byte offset inside a row of the wave's block (4 x lane)
S_WB, S_WP, S_SB3 = 96, 98, 100                # the wave's block; the 4 KB page of it an instruction addresses; the third scratch column


def text_of_list(ins):
    a = new_asm(); a.ins = ins
    if not TWO_WAVE:
        return a.text()
    out, page, in_loads = [], None, False
    for t in ins:
        if t[0] == "hook":
            continue
        if t[0] in ("aread", "awrite"):
            areg = t[2] if t[0] == "aread" else t[1]
            vreg = t[1] if t[0] == "aread" else t[2]
            byte = areg * 256
            if byte // 4096 != page:
                page = byte // 4096
                out += ["s_add_u32 s%d, s%d, %d" % (S_WP, S_WB, page * 4096), "s_addc_u32 s%d, s%d, 0" % (S_WP + 1, S_WB + 1)]
            if t[0] == "aread":
                if not in_loads:
                    out.append("s_waitcnt vmcnt(0)")                  # earlier stores of this wave have landed
                    in_loads = True
                out.append("global_load_dword v%d, v%d, s[%d:%d] offset:%d" % (vreg, V_WOFF, S_WP, S_WP + 1, byte % 4096))
            else:
                if in_loads:
                    out.append("s_waitcnt vmcnt(0)")
                    in_loads = False
                out.append("global_store_dword v%d, v%d, s[%d:%d] offset:%d" % (V_WOFF, vreg, S_WP, S_WP + 1, byte % 4096))
            continue
        if in_loads:
            out.append("s_waitcnt vmcnt(0)")
            in_loads = False
        if t[0] == "call":
            page = None                                                # a subroutine may move the page pointer
        out += a.text_of(t).split("\n")
    if in_loads:
        out.append("s_waitcnt vmcnt(0)")
    return out


def kernel_text():
    """operands: %0 (output, v) flag: 1 = recompute this lane with the complete formulas; %1 mapped points M (s pair); %2 row stride of M in bytes (s);
    %3 output H (s pair); %4 its row stride in bytes (s); %5 scratch columns (s pair: two per lane, 24 rows each); %6 their row stride in bytes (s);
    %7 lane index i (v: this lane reads columns 2 i, 2 i + 1 of M and owns column i of H and of the scratch); %8 LDS address of three slots (s)"""
    prep, dbl, add, psi = build_level2()
    T = []
    T += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK), "s_mov_b32 s%d, 0x%x" % (S_RECIP, RECIP)]
    T += ["s_mov_b64 s[%d:%d], %%1" % (S_MB, S_MB + 1), "s_mov_b32 s%d, %%2" % S_MSTR, "s_mov_b64 s[%d:%d], %%3" % (S_HB, S_HB + 1), "s_mov_b32 s%d, %%4" % S_HSTR,
          "s_mov_b64 s[%d:%d], %%5" % (S_SB, S_SB + 1), "s_mov_b32 s%d, %%6" % S_SSTR]
    # second scratch column = + 24 rows
    T += ["s_mul_hi_u32 s%d, s%d, 24" % (S_T, S_SSTR), "s_mul_i32 s%d, s%d, 24" % (S_T + 1, S_SSTR),
          "s_add_u32 s%d, s%d, s%d" % (S_SB + 2, S_SB, S_T + 1), "s_addc_u32 s%d, s%d, s%d" % (S_SB + 3, S_SB + 1, S_T)]
    T += ["v_lshlrev_b32_e64 v%d, 5, %%7" % V_OFFM, "v_lshlrev_b32_e64 v%d, 4, %%7" % V_OFFH]
    T += ["v_mbcnt_lo_u32_b32 v%d, -1, 0" % TMP, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (TMP, TMP)]
    if TWO_WAVE:
        # %9: this wave's block of 252 rows x 256 bytes (s pair); the third scratch column = + 48 rows
        T += ["v_lshlrev_b32_e64 v%d, 2, v%d" % (V_WOFF, TMP), "s_mov_b64 s[%d:%d], %%9" % (S_WB, S_WB + 1),
              "s_add_u32 s%d, s%d, s%d" % (S_SB3, S_SB + 2, S_T + 1), "s_addc_u32 s%d, s%d, s%d" % (S_SB3 + 1, S_SB + 3, S_T)]
    T += ["v_lshlrev_b32_e64 v%d, 4, v%d" % (TMP, TMP), "v_add_u32_e64 v%d, %%8, v%d" % (V_LDS, TMP)]
    for name, sreg in SUB_ADDR.items():
        T += ["s_getpc_b64 s[%d:%d]" % (sreg, sreg + 1), ".Lcc_p%s%%=:" % name,
              "s_add_u32 s%d, s%d, (.Lcc_%s%%=-.Lcc_p%s%%=)&4294967295" % (sreg, sreg, name, name),
              "s_addc_u32 s%d, s%d, (.Lcc_%s%%=-.Lcc_p%s%%=)>>32" % (sreg + 1, sreg + 1, name, name)]
    T += ["s_branch .Lcc_main%="]
    for name, fn in LEAVES.items():
        T += [".Lcc_%s%%=:" % name] + text_of_list(fn()) + ["s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1)]
    # the chain first (hot): its loop, the doubling body, then ADD right behind it; PREP and PSI (cold) after
    T += [".Lcc_CHAIN%=:"] + text_of_list(chain_ins())
    T += ["s_mov_b32 s%d, 0x%x" % (S_XA, X_ABS & 0xffffffff), "s_mov_b32 s%d, 0x%x" % (S_XA + 1, X_ABS >> 32), "s_mov_b32 s%d, 62" % S_I]
    T += [".Lcc_loop%=:"] + text_of_list(dbl)
    T += ["s_lshr_b64 s[%d:%d], s[%d:%d], s%d" % (S_T, S_T + 1, S_XA, S_XA + 1, S_I), "s_bitcmp1_b32 s%d, 0" % S_T, "s_cbranch_scc0 .Lcc_next%=",
          "s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_RET2, S_RET2 + 1, S_ADD, S_ADD + 1)]
    T += [".Lcc_next%=:", "s_sub_u32 s%d, s%d, 1" % (S_I, S_I), "s_cmp_ge_i32 s%d, 0" % S_I, "s_cbranch_scc1 .Lcc_loop%=", "s_setpc_b64 s[%d:%d]" % (S_RET3, S_RET3 + 1)]
    T += [".Lcc_ADD%=:"] + text_of_list(add) + ["s_setpc_b64 s[%d:%d]" % (S_RET2, S_RET2 + 1)]
    T += [".Lcc_PREP%=:"] + text_of_list(prep) + ["s_setpc_b64 s[%d:%d]" % (S_RET2, S_RET2 + 1)]
    T += [".Lcc_PSI%=:"] + text_of_list(psi) + ["s_setpc_b64 s[%d:%d]" % (S_RET2, S_RET2 + 1)]
    T += [".Lcc_main%=:"] + text_of_list(main_program(dbl))
    T += ["v_mov_b32_e64 %%0, v%d" % V_FLAG]
    return T


def clobbers():
    if TWO_WAVE:
        c = ["v%d" % i for i in range(CLOBBER_V)] + ["v%d" % V_WOFF] + ["s%d" % i for i in range(36, 102)] + ["vcc", "scc", "memory"]
    else:
        c = ["v%d" % i for i in range(CLOBBER_V)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in range(*CLOBBER_S)] + ["vcc", "scc", "memory"]
    return ", ".join('"%s"' % x for x in c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("-o", "--out")
    ap.add_argument("--two-wave", action="store_true", help="the 256-register form (no AGPRs, no LDS): BLS_CLEAR2_ASM_BODY")
    a = ap.parse_args()
    global TWO_WAVE
    TWO_WAVE = a.two_wave
    if a.selftest:
        selftest(walk=not a.quick)
        return
    lines = [l if l.startswith(".L") else "\\t" + l for l in kernel_text()]
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_clear_asm.py -- do not edit.\n"
           "// operands: %0 flag out (v), %1 M (s pair), %2 M row stride bytes (s), %3 H (s pair), %4 H row stride bytes (s), %5 scratch columns (s pair), %6 their row stride (s), %7 lane index (v), %8 LDS address of three slots (s)\n"
           + ("// two-wave form: %9 this wave's block of 252 rows x 256 bytes (s pair); %8 unused\n" if TWO_WAVE else "") +
           "#define BLS_CLEAR%s_ASM_BODY \\\n" % ("2" if TWO_WAVE else "") + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_CLEAR%s_ASM_CLOBBERS " % ("2" if TWO_WAVE else "") + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
