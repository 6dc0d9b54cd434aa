#!/usr/bin/env python3
"""Generates the hand-allocated gfx950 loop of k_lines (csrc/kernels.hip) as ONE inline-asm statement: the 63 doubling steps and 5
addition steps of the Miller loop of one pair (P in G1, Q in G2) per lane, with the 68 sparse line functions stored step-major.

What it computes is pairing.hpp's miller_lines / miller_dbl_step / miller_add_step (same formulas, same carry and reduction points,
bounds re-checked here while generating, asmlib.Builder); only the bookkeeping differs:
  * the walking point T = (X, Y, Z), B = Y^2 and E = 12 xi Z^2 live in fixed VGPR blocks, Q and the P-side factors (Z^3, X Z, -3 X Z, Y of
    P) in AGPRs; nothing is ever spilled;
  * four multiplier SUBROUTINES read their operands from fixed slots SA, SB (an Fp2 each) and K and leave their results in slots:
      SQR   SA^2          -> (R0, K)        two Fp products (a0 + a1)(a0 - a1), (2 a0) a1
      MUL   SA * SB       -> (R0, SA1)      two lazily reduced dot products (fp_dot2)
      MULK  SA * K (Fp)   -> (R0, K)        two Fp products
      DOT2  SA0 SB0 + SA1 SB1 -> SA0        one lazily reduced dot product
    a call is one s_swappc_b64, a return one s_setpc_b64: no argument protocol, no callee-entry wait;
  * derived operands (sums, differences, small multiples, carried forms) are computed straight INTO the slots, so a plain register copy
    is needed only where a stored value enters a product unchanged (~350 copies per doubling step against ~11 300 multiplier
    instructions and ~1 400 of formula glue);
  * E = reduce(12 xi C) folds the factor 12 into fp_reduce's own multiply-add chain (no dbl / carry / dbl / triple in front);
  * the 24 line stores of a step are issued from the result registers and never waited for.
Hot code: the four subroutines (31 KB) + the doubling step's glue (~12 KB); the addition step (5 of 68 steps) lies outside the loop's
fall-through path.

`--selftest` runs the generated instruction lists for one lane in asmlib's interpreter against big-integer arithmetic: single steps
with every intermediate bound checked, and the whole 68-step walk (tests/test_asm_loops.py).
"""
import argparse
import random
import sys

from asmlib import (Asm, Builder, F2, Fp, Fp2, LB, MASK, Machine, N0, NL, ONE, P, PL, RECIP, X_ABS, blk, blk2, check_limbs, get, limbs_of, mmul,
                    put, rand_fp)

# ---- register plan ------------------------------------------------------------------------------------------------------------------
SA, SB = blk2(0), blk2(28)                       # operand slots (an Fp2 each)
R0, K = blk(56), blk(70)                         # result block / Fp operand of MULK (also SQR's scratch and second result)
M_REGS = [84 + i for i in range(NL)]             # Montgomery quotient digits
FREE = [98 + 14 * i for i in range(10)]          # ten value blocks: v98 .. v237
X, Y, Z, B, E = blk2(FREE[0]), blk2(FREE[2]), blk2(FREE[4]), blk2(FREE[6]), blk2(FREE[8])
ACC, TMP, TMP2, V_OFF, V_LDS = 238, 240, 241, 242, 243          # v[238:239] column accumulator; scratch; byte offset of this lane's pair; LDS address
CLOBBER_V = 250
# AGPRs
AQX, AQY, AQZ = blk2(0, True), blk2(28, True), blk2(56, True)                                   # Q (homogeneous, reduced)
AZ3, AXZ, ANXZ3, AYP = blk(84, True), blk(98, True), blk(112, True), blk(126, True)            # Z^3, X Z, -3 X Z, Y of P
AT = [blk2(168 + 28 * i, True) for i in range(3)]                                               # addition step: parked intermediates (a168 .. a251)
# SGPRs
S_P, S_N0, S_MASK, S_RECIP = 36, 50, 51, 52      # s36..s49 limbs of p
S_SQR, S_MUL, S_MULK, S_DOT2, S_RET = 54, 56, 58, 60, 62      # subroutine addresses (pairs), return address (pair)
S_LINE, S_LP, S_STR, S_STR8, S_STEP = 64, 66, 68, 69, 70      # step base (pair), row pointer (pair), row stride in bytes, 8 rows, 24 rows
S_I, S_XA, S_T, S_EXEC = 71, 72, 74, 76          # bit index; |x| (pair); temporaries (pair); saved exec (pair)
CLOBBER_S = (36, 78)


def new_asm():
    return Asm(ACC, S_P, S_N0, S_MASK, S_RECIP)


def builder(a):
    return Builder(a, M_REGS, TMP, TMP2)


def f2(b, d, c0, c1):
    return Fp2(c0, c1)


# ---- subroutines ----------------------------------------------------------------------------------------------------------------
def sub_sqr():
    """SA^2 -> (R0, K).  Operands: |limb| within one unit (carried), vb(a0) vb(a1) such that the two products fit (checked by the caller)."""
    a = new_asm(); b = builder(a)
    x0, x1 = SA.c0.like(16, 1), SA.c1.like(16, 1)             # the widest operand a caller passes (Z: |v| < 8 p; Y + Z: 10 p; here 16 p)
    s = b.add_nc(R0, x0, x1)
    d = b.sub_nc(K, x0, x1)
    b.dot([(s, d)], R0)
    d2 = b.shl(K, x0, 1)
    b.dot([(d2, x1)], K)
    return a.ins


def sub_mul():
    """SA * SB -> (R0, SA1): re = a0 b0 - a1 b1, im = a0 b1 + a1 b0, one reduction each."""
    a = new_asm(); b = builder(a)
    a0, a1, b0, b1 = SA.c0.like(16, 2), SA.c1.like(16, 2), SB.c0.like(16, 2), SB.c1.like(16, 2)      # widest case checked at the call sites
    na1 = b.neg(R0, a1)
    b.dot_body([(a0, b0), (na1, b1)], R0)
    b.dot_body([(a0, b1), (a1, b0)], SA.c1)
    return a.ins


def sub_mulk():
    """(SA0 K, SA1 K) -> (R0, K)"""
    a = new_asm(); b = builder(a)
    b.dot_body([(SA.c0, K)], R0)
    b.dot_body([(SA.c1, K)], K)
    return a.ins


def sub_dot2():
    """SA0 SB0 + SA1 SB1 -> SA0"""
    a = new_asm(); b = builder(a)
    b.dot_body([(SA.c0, SB.c0), (SA.c1, SB.c1)], SA.c0)
    return a.ins


SUBS = {"SQR": (S_SQR, sub_sqr), "MUL": (S_MUL, sub_mul), "MULK": (S_MULK, sub_mulk), "DOT2": (S_DOT2, sub_dot2)}


class Steps:
    """the doubling and the addition step as instruction lists (with bounds propagated from one step to the next)"""

    def __init__(self):
        self.a = new_asm()
        self.b = builder(self.a)
        self.store_hook = None          # interpreter only: called with (coefficient index, Fp2 in registers)

    # -- calls with their preconditions
    def call(self, name):
        self.a.e("call", name, S_RET, SUBS[name][0])

    def SQR(self, x):
        """x: Fp2 whose bounds describe what sits in SA"""
        b0 = max(x.c0.vb, x.c1.vb)
        assert x.c0.lb <= 1 and x.c1.lb <= 1 and (2 * b0) * (2 * b0) <= 2048 and b0 <= 16, "SQR operand bounds"
        self.call("SQR")
        return Fp2(R0.like(2, 0), K.like(2, 0))

    def MUL(self, x, y):
        pairs_re = [(x.c0, y.c0), (x.c1, y.c1)]
        from asmlib import dot_bounds_ok
        assert dot_bounds_ok(pairs_re) and dot_bounds_ok([(x.c0, y.c1), (x.c1, y.c0)]), "MUL operand bounds"
        self.call("MUL")
        return Fp2(R0.like(2, 0), SA.c1.like(2, 0))

    def MULK(self, x, k):
        from asmlib import dot_bounds_ok
        assert dot_bounds_ok([(x.c0, k)]) and dot_bounds_ok([(x.c1, k)]), "MULK operand bounds"
        self.call("MULK")
        return Fp2(R0.like(2, 0), K.like(2, 0))

    def DOT2(self, x0, y0, x1, y1):
        from asmlib import dot_bounds_ok
        assert dot_bounds_ok([(x0, y0), (x1, y1)]), "DOT2 operand bounds"
        self.call("DOT2")
        return SA.c0.like(2, 0)

    # -- Fp2 helpers on the builder
    def mov2(self, d, x):
        return Fp2(self.b.mov(d.c0, x.c0), self.b.mov(d.c1, x.c1))

    def carry2(self, d, x):
        return Fp2(self.b.carry(d.c0, x.c0), self.b.carry(d.c1, x.c1))

    def add2(self, d, x, y):
        return Fp2(self.b.add_nc(d.c0, x.c0, y.c0), self.b.add_nc(d.c1, x.c1, y.c1))

    def sub2(self, d, x, y):
        return Fp2(self.b.sub_nc(d.c0, x.c0, y.c0), self.b.sub_nc(d.c1, x.c1, y.c1))

    def store_line(self, coeff, v):
        """line coefficient `coeff` (0..2) of this step <- the Fp2 v (registers): 8 rows of the step-major line store, issued and never
        waited for.  Row pointer = step base + coeff * 8 rows."""
        a = self.a
        a.e("raw", "s_mov_b64 s[%d:%d], s[%d:%d]" % (S_LP, S_LP + 1, S_LINE, S_LINE + 1))
        for _ in range(coeff):
            a.e("raw", "s_add_u32 s%d, s%d, s%d" % (S_LP, S_LP, S_STR8))
            a.e("raw", "s_addc_u32 s%d, s%d, 0" % (S_LP + 1, S_LP + 1))
        for part in (v.c0, v.c1):
            for q in range(4):
                r = part.r[4 * q]
                if q < 3:
                    a.e("raw", "global_store_dwordx4 v%d, v[%d:%d], s[%d:%d]" % (V_OFF, r, r + 3, S_LP, S_LP + 1))
                else:
                    a.e("raw", "global_store_dwordx2 v%d, v[%d:%d], s[%d:%d]" % (V_OFF, r, r + 1, S_LP, S_LP + 1))
                a.e("raw", "s_add_u32 s%d, s%d, s%d" % (S_LP, S_LP, S_STR))
                a.e("raw", "s_addc_u32 s%d, s%d, 0" % (S_LP + 1, S_LP + 1))
        a.e("raw", "s_nop 1")                                   # store data registers are rewritten soon: keep the two wait states of the >64-bit store hazard explicit
        if self.store_hook:
            hook, c0, c1 = self.store_hook, v.c0, v.c1
            a.e("hook", lambda mach, coeff=coeff, c0=c0, c1=c1: hook(mach, coeff, c0, c1))

    def next_step(self):
        a = self.a
        a.e("raw", "s_add_u32 s%d, s%d, s%d" % (S_LINE, S_LINE, S_STEP))
        a.e("raw", "s_addc_u32 s%d, s%d, 0" % (S_LINE + 1, S_LINE + 1))

    # -- the doubling step: T <- 2T, tangent line at T evaluated at P (pairing.hpp miller_dbl_step_m)
    def dbl(self, T):
        b = self.b
        Xv, Yv, Zv = T
        # B = Y^2
        self.mov2(SA, Yv)
        Bv = self.mov2(B, self.SQR(Yv))
        # S1 = (Y + Z)^2 before C = Z^2, so that C is consumed from the result registers (no parking): t = S1 - B waits in SB
        yz = self.carry2(SA, self.add2(SA, Yv, Zv))
        S1 = self.SQR(yz)
        t = self.sub2(SB, S1, Bv)
        # C = Z^2;  E = reduce(12 xi C);  H = 2 Y Z = S1 - B - C, limb-wise (two units: both products it enters take that)
        self.mov2(SA, Zv)
        Cv = self.SQR(Zv)
        Ha = self.sub2(SA, t, Cv)
        u = b.sub_nc(SB.c0, Cv.c0, Cv.c1)
        v = b.add_nc(SB.c1, Cv.c0, Cv.c1)
        Ev = Fp2(b.reduce(E.c0, u, 12), b.reduce(E.c1, v, 12))
        # line coefficient 2: H * y_P
        kv = b.mov(K, AYP.like(1, 1))
        self.store_line(2, self.MULK(Ha, kv))
        # Z3 = 4 B H  (SA = H, SB <- B)
        Bb = self.mov2(SB, Bv)
        BH = self.MUL(Ha, Bb)
        Z3 = Fp2(b.carry(Z.c0, b.shl(Z.c0, BH.c0, 2)), b.carry(Z.c1, b.shl(Z.c1, BH.c1, 2)))
        # 2 X Y = (X + Y)^2 - X^2 - B
        xy = self.carry2(SA, self.add2(SA, Xv, Yv))
        S2 = self.SQR(xy)
        t = self.sub2(SB, S2, Bv)
        self.mov2(SA, Xv)
        X2 = self.SQR(Xv)
        XY2 = self.sub2(SB, t, X2)                              # limb-wise, two units: (B - 3 E) below is carried, the product takes 2 x 1
        # line coefficient 1: X^2 * (-3 X Z of P)
        x2a = self.mov2(SA, X2)
        kv = b.mov(K, ANXZ3.like(3, 1))
        self.store_line(1, self.MULK(x2a, kv))
        # X3 = 2 X Y (B - 3 E)
        F0, F1 = b.mul3(SA.c0, Ev.c0), b.mul3(SA.c1, Ev.c1)
        bf = self.carry2(SA, Fp2(b.sub_nc(SA.c0, Bv.c0, F0), b.sub_nc(SA.c1, Bv.c1, F1)))
        X3 = self.mov2(X, self.MUL(bf, XY2))
        # line coefficient 0: (B - E) * z3_P
        be = self.sub2(SA, Bv, Ev)
        kv = b.mov(K, AZ3.like(2, 1))
        self.store_line(0, self.MULK(be, kv))
        # Y3 = (B + 3 E)^2 - 12 E^2 as two lazily reduced dot products (pairing.hpp fp2_sqr_minus_12sqr)
        a0 = b.carry(R0, b.add_nc(R0, Bv.c0, b.mul3(R0, Ev.c0)))
        a1 = b.carry(SB.c0, b.add_nc(SB.c0, Bv.c1, b.mul3(SB.c0, Ev.c1)))
        # imaginary part first (its second operand IS a1): [2 a0] a1 - [8 e0] [3 e1]
        x0 = b.shl(SA.c0, a0, 1)
        e8 = b.shl(SA.c1, b.carry(SA.c1, b.shl(SA.c1, Ev.c0, 2)), 1)
        x1 = b.neg(SA.c1, e8)
        y1 = b.carry(SB.c1, b.mul3(SB.c1, Ev.c1))
        Y3im = b.mov(Y.c1, self.DOT2(x0, a1, x1, y1))
        # real part: (a0 + a1)(a0 - a1) - [4 (e0 + e1)] [3 (e0 - e1)]
        x0 = b.add_nc(SA.c0, a0, a1)
        y0 = b.sub_nc(SB.c0, a0, a1)
        e4s = b.shl(SA.c1, b.carry(SA.c1, b.shl(SA.c1, b.add_nc(SA.c1, Ev.c0, Ev.c1), 1)), 1)
        x1 = b.neg(SA.c1, e4s)
        ed = b.sub_nc(SB.c1, Ev.c0, Ev.c1)
        y1 = b.carry(SB.c1, b.mul3(SB.c1, ed))
        Y3re = b.mov(Y.c0, self.DOT2(x0, y0, x1, y1))
        self.next_step()
        return (X3, Fp2(Y3re, Y3im), Z3)

    # -- the addition step: T <- T + Q, chord through T and Q evaluated at P (pairing.hpp miller_add_step); runs 5 times in 68 steps
    def add(self, T):
        b = self.b
        Xv, Yv, Zv = T
        Q = (Fp2(AQX.c0.like(1, 0), AQX.c1.like(1, 0)), Fp2(AQY.c0.like(1, 0), AQY.c1.like(1, 0)), Fp2(AQZ.c0.like(1, 0), AQZ.c1.like(1, 0)))
        W0, W1 = B, E                                          # the doubling step's B and E blocks are free here
        A0, A1, A2 = AT
        qz = self.mov2(SB, Q[2])
        self.mov2(SA, Yv)
        Y1Z2 = self.mov2(A0, self.MUL(Yv, qz))                 # parked: needed at the very end
        self.mov2(SA, Xv)
        X1Z2 = self.mov2(W0, self.MUL(Xv, qz))
        self.mov2(SA, Zv)
        Z1Z2 = self.mov2(A1, self.MUL(Zv, qz))
        tz = self.mov2(SB, Zv)
        qy = self.mov2(SA, Q[1])
        Y2Z1 = self.MUL(qy, tz)
        y1z2 = self.mov2(SB, Y1Z2)                             # from the AGPR copy (SB is free: Z1 is read again below from its home)
        u = self.carry2(W1, self.sub2(W1, Y2Z1, y1z2))
        tz = self.mov2(SB, Zv)
        qx = self.mov2(SA, Q[0])
        X2Z1 = self.MUL(qx, tz)
        v = self.carry2(Z, self.sub2(Z, X2Z1, X1Z2))           # Z1 is dead now: v takes its block
        # c0 = u X2 - v Y2 first (needs u, v and Q only), c1 = u Z2, c2 = v Z2: the three line coefficients
        qx = self.mov2(SB, Q[0])
        self.mov2(SA, u)
        uX2 = self.mov2(A2, self.MUL(u, qx))
        qy = self.mov2(SB, Q[1])
        self.mov2(SA, v)
        vY2 = self.MUL(v, qy)
        ux = self.mov2(SB, uX2)
        c0 = self.carry2(SA, self.sub2(SA, ux, vY2))
        kv = b.mov(K, AZ3.like(2, 1))
        self.store_line(0, self.MULK(c0, kv))
        qz = self.mov2(SB, Q[2])
        self.mov2(SA, u)
        c1 = self.MUL(u, qz)
        c1a = self.mov2(SA, c1)                                # (R0, SA1) -> SA
        kv = b.mov(K, AXZ.like(1, 1))
        l1 = self.MULK(c1a, kv)
        n1 = Fp2(b.neg(SA.c0, l1.c0), b.neg(SA.c1, l1.c1))    # the line's v coefficient is -(c1 x z)
        self.store_line(1, n1)
        self.mov2(SA, v)
        c2 = self.MUL(v, qz)
        c2a = self.mov2(SA, c2)
        kv = b.mov(K, AYP.like(1, 1))
        self.store_line(2, self.MULK(c2a, kv))
        # uu, vv, vvv, R
        self.mov2(SA, u)
        uu = self.mov2(X, self.SQR(u))                         # X1 is dead (X1Z2 is in W0): uu takes its block
        self.mov2(SA, v)
        vv = self.mov2(Y, self.SQR(v))                         # Y1 dead (Y1Z2 parked)
        self.mov2(SB, vv)
        self.mov2(SA, v)
        vvv = self.mov2(A2, self.MUL(v, vv))                   # parked (three more uses)
        # SB still holds vv
        self.mov2(SA, X1Z2)
        Rr = self.mov2(W0, self.MUL(X1Z2, vv))                 # R replaces X1Z2
        z1z2 = self.mov2(SB, Z1Z2)
        self.mov2(SA, uu)
        uuZ = self.MUL(uu, z1z2)
        vvvb = self.mov2(SB, vvv)
        t = self.sub2(SA, uuZ, vvvb)
        Aa = self.carry2(SA, Fp2(b.sub_nc(SA.c0, t.c0, b.shl(X.c0, Rr.c0, 1)), b.sub_nc(SA.c1, t.c1, b.shl(X.c1, Rr.c1, 1))))      # uu's block is scratch now
        Aa = self.mov2(Y, Aa)                                  # vv is dead: A takes its block
        vb_ = self.mov2(SB, v)
        X3 = self.MUL(Aa, vb_)                                 # SA = A
        X3 = self.mov2(X, X3)
        ra = self.sub2(SA, Rr, Aa)                             # R - A, limb-wise (two units: the product takes it)
        ub = self.mov2(SB, u)
        uRA = self.mov2(W0, self.MUL(ra, ub))                  # R is dead
        y1z2 = self.mov2(SB, Y1Z2)
        self.mov2(SA, vvv)
        vY = self.MUL(vvv, y1z2)
        Y3 = Fp2(b.reduce(Y.c0, b.sub_nc(Y.c0, uRA.c0, vY.c0)), b.reduce(Y.c1, b.sub_nc(Y.c1, uRA.c1, vY.c1)))
        z1z2 = self.mov2(SB, Z1Z2)
        self.mov2(SA, vvv)
        Z3 = self.mov2(Z, self.MUL(vvv, z1z2))
        self.next_step()
        return (X3, Y3, Z3)


# ---- reference model (big integers, Montgomery images) ---------------------------------------------------------------------------------
def ref_dbl(T, pre):
    Xr, Yr, Zr = T
    z3p, xzp, nxz3p, yp = pre
    Bq, Cq, X2q = Yr.sqr(), Zr.sqr(), Xr.sqr()
    Eq = Cq.xi() * 12
    Hq = (Yr + Zr).sqr() - Bq - Cq
    XY2 = (Xr + Yr).sqr() - X2q - Bq
    x3 = XY2 * (Bq - Eq * 3)
    a = Bq + Eq * 3
    y3 = a.sqr() - Eq.sqr() * 12
    z3 = (Bq * Hq) * 4
    line = ((Bq - Eq).mulfp(z3p), X2q.mulfp(nxz3p), Hq.mulfp(yp))
    return (x3, y3, z3), line


def ref_add(T, Q, pre):
    Xr, Yr, Zr = T
    z3p, xzp, nxz3p, yp = pre
    Y1Z2, X1Z2, Z1Z2 = Yr * Q[2], Xr * Q[2], Zr * Q[2]
    u = Q[1] * Zr - Y1Z2
    v = Q[0] * Zr - X1Z2
    uu, vv = u.sqr(), v.sqr()
    vvv = v * vv
    Rr = vv * X1Z2
    Aa = uu * Z1Z2 - vvv - Rr * 2
    x3 = v * Aa
    y3 = u * (Rr - Aa) - vvv * Y1Z2
    z3 = vvv * Z1Z2
    c0 = u * Q[0] - v * Q[1]
    c1, c2 = u * Q[2], v * Q[2]
    return (x3, y3, z3), (c0.mulfp(z3p), -(c1.mulfp(xzp)), c2.mulfp(yp))


# ---- self-test ------------------------------------------------------------------------------------------------------------------------
def selftest(walk=True, seed=11):
    rnd = random.Random(seed)
    subs = {name: fn() for name, (_, fn) in SUBS.items()}
    proto = new_asm()
    mach = Machine(proto, subs)
    lines_out = []

    def hook(m, coeff, c0, c1):
        check_limbs(m, c0, c0.lb)            # canonical limbs, or (the addition step's middle coefficient) a limb-wise negation of them
        check_limbs(m, c1, c1.lb)
        assert c0.lb <= 1 and c1.lb <= 1 and c0.vb <= 2 and c1.vb <= 2      # what k_lineprod's loop takes (gen_lineprod_asm.py)
        lines_out.append((coeff, F2(get(m, c0), get(m, c1))))

    def load_state(T_regs, T_vals):
        for reg, val in zip(T_regs, T_vals):
            put(mach, reg.c0, limbs_of(val.c0))
            put(mach, reg.c1, limbs_of(val.c1))

    # inputs: Q reduced (|v| < 0.51 p: canonical limbs of a residue or limb-wise negated), P-side factors as g1_precompute leaves them
    def rand_f2():
        return F2(rnd.randrange(P), rnd.randrange(P))

    Qv = (rand_f2(), rand_f2(), rand_f2())
    pre = tuple(rnd.randrange(P) for _ in range(4))
    for reg, val in zip((AQX, AQY, AQZ), Qv):
        put(mach, reg.c0, limbs_of(val.c0)); put(mach, reg.c1, limbs_of(val.c1))
    for reg, val in zip((AZ3, AXZ, ANXZ3, AYP), pre):
        # -3 X Z is stored limb-wise negated by g1_precompute (fp_neg of a carried value): exercise signed limbs there
        ls = limbs_of(val) if reg is not ANXZ3 else [(-l) & 0xffffffff for l in limbs_of((P - val) % P)]
        put(mach, reg, ls)
    st = Steps()
    st.store_hook = hook
    T0 = (Fp2(X.c0.like(1, 0), X.c1.like(1, 0)), Fp2(Y.c0.like(1, 0), Y.c1.like(1, 0)), Fp2(Z.c0.like(1, 0), Z.c1.like(1, 0)))
    # ---- generate: first doubling from the initial bounds, then the steady-state doubling, the addition, and a doubling behind an addition
    T1 = st.dbl(T0); first_dbl = st.a.ins; st.a.ins = []
    T2 = st.dbl(T1); dbl_ins = st.a.ins; st.a.ins = []
    T3 = st.dbl(T2); st.a.ins = []
    for p_, q_ in zip(T2, T3):                               # the doubling step's output bounds are a fixed point: ONE loop body serves all 63 steps
        assert (p_.c0.vb, p_.c0.lb, p_.c1.vb, p_.c1.lb) == (q_.c0.vb, q_.c0.lb, q_.c1.vb, q_.c1.lb)
    for p_, q_ in zip(T1, T2):
        assert q_.c0.vb <= max(p_.c0.vb, q_.c0.vb)
    Ta = st.add(T2); add_ins = st.a.ins; st.a.ins = []
    Tb = st.dbl(Ta); st.a.ins = []
    for p_, q_ in zip(Tb, T2):                               # ... including the one behind an addition step (its inputs are no wider)
        assert p_.c0.vb <= q_.c0.vb and p_.c0.lb <= q_.c0.lb
    for p_, q_ in zip(Ta, T2):
        assert p_.c0.vb <= q_.c0.vb and p_.c1.vb <= q_.c1.vb and p_.c0.lb <= q_.c0.lb, "the loop body (generated for the doubling's output bounds) also takes the addition's output"
    for p_, q_ in zip(T0, T2):
        assert p_.c0.vb <= q_.c0.vb and p_.c0.lb <= q_.c0.lb, "... and the initial point"

    def check_T(Tregs, Tref, what):
        for reg, ref in zip(Tregs, Tref):
            got = F2(get(mach, reg.c0), get(mach, reg.c1))
            assert got == ref, (what, "coordinate mismatch")
            for c in (reg.c0, reg.c1):
                check_limbs(mach, c, c.lb)
                assert abs(get(mach, c)) <= c.vb * P, (what, "value bound")

    def check_lines(ref_line, what):
        got = dict(lines_out)
        assert len(lines_out) == 3 and all(got[i] == ref_line[i] for i in range(3)), (what, "line mismatch")
        del lines_out[:]

    Tref = Qv
    load_state((X, Y, Z), Tref)
    # single steps
    mach.run(dbl_ins)
    Tref, line = ref_dbl(Tref, pre)
    check_T(T2, Tref, "dbl"); check_lines(line, "dbl")
    n_dbl = dict(mach.count)
    mach.run(add_ins)
    Tref, line = ref_add(Tref, Qv, pre)
    check_T(Ta, Tref, "add"); check_lines(line, "add")
    n_add = {k: mach.count[k] - n_dbl[k] for k in n_dbl}
    if walk:
        Tref = Qv
        load_state((X, Y, Z), Tref)
        steps = 0
        for bit in range(62, -1, -1):
            mach.run(dbl_ins)
            Tref, line = ref_dbl(Tref, pre)
            check_T(T2, Tref, "walk dbl %d" % bit); check_lines(line, "walk dbl")
            steps += 1
            if (X_ABS >> bit) & 1:
                mach.run(add_ins)
                Tref, line = ref_add(Tref, Qv, pre)
                check_T(Ta, Tref, "walk add %d" % bit); check_lines(line, "walk add")
                steps += 1
        assert steps == 68
    print("gen_lines_asm selftest ok: doubling step %d VALU instructions (%d multiply-adds, %.1f %%, %d calls), addition step %d (%d, %d calls); "
          "subroutines %s instructions"
          % (n_dbl["valu"], n_dbl["mad"], 100.0 * n_dbl["mad"] / n_dbl["valu"], n_dbl["calls"], n_add["valu"], n_add["mad"], n_add["calls"],
             {k: len(v) for k, v in subs.items()}))
    return dbl_ins, add_ins, subs


# ---- text ---------------------------------------------------------------------------------------------------------------------------
def lds_fetch(dst_blk2, lds_vreg, slot_off):
    """one Fp2 (28 words per lane, 7 groups of 64 lanes x 16 bytes: fp2_lds_put's layout) -> 28 consecutive VGPRs"""
    base = dst_blk2.c0.r[0]
    assert dst_blk2.c1.r[0] == base + NL
    return ["ds_read_b128 v[%d:%d], v%d offset:%d" % (base + 4 * q, base + 4 * q + 3, lds_vreg, slot_off + 1024 * q) for q in range(7)]


def kernel_text():
    """operands: %0 lines base (s pair), %1 row stride in bytes (s), %2 byte offset of this lane's pair = 16 * pair index (v), %3 skip flag (v: != 0 -> the
    lane stores nothing), %4 LDS address of the four hand-over slots Q.x | Q.y | Q.z | (z3, -3xz) (s), %5 LDS address of the fifth (xz, y) (s)"""
    st = Steps()
    T0 = (Fp2(X.c0.like(8, 1), X.c1.like(8, 1)), Fp2(Y.c0.like(2, 0), Y.c1.like(2, 0)), Fp2(Z.c0.like(8, 1), Z.c1.like(8, 1)))
    T0 = (Fp2(X.c0.like(2, 0), X.c1.like(2, 0)), T0[1], T0[2])
    T1 = st.dbl(T0); dbl_ins = st.a.text(); st.a.ins = []
    for p_, q_ in zip(T1, T0):
        assert (p_.c0.vb, p_.c0.lb) == (q_.c0.vb, q_.c0.lb), "loop body generated at its fixed point"
    Ta = st.add(T1); add_ins = st.a.text(); st.a.ins = []
    T = []
    T += ["s_mov_b64 s[%d:%d], exec" % (S_EXEC, S_EXEC + 1), "v_cmp_eq_u32_e64 vcc, 0, %3", "s_and_b64 exec, exec, vcc", "s_cbranch_execz .Lml_end%="]
    T += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK), "s_mov_b32 s%d, 0x%x" % (S_RECIP, RECIP)]
    T += ["s_mov_b64 s[%d:%d], %%0" % (S_LINE, S_LINE + 1), "s_mov_b32 s%d, %%1" % S_STR, "s_lshl_b32 s%d, %%1, 3" % S_STR8, "s_mul_i32 s%d, %%1, 24" % S_STEP]
    T += ["v_mov_b32_e64 v%d, %%2" % V_OFF]
    T += ["v_mbcnt_lo_u32_b32 v%d, -1, 0" % TMP, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (TMP, TMP), "v_lshlrev_b32_e64 v%d, 4, v%d" % (TMP, TMP)]
    # hand-over: T = Q into the homes, Q and the P-side factors into their AGPR homes
    T += ["v_add_u32_e64 v%d, %%4, v%d" % (V_LDS, TMP)]
    for s_, dst in enumerate((X, Y, Z, SA)):
        T += lds_fetch(dst, V_LDS, 7168 * s_)
    T += ["v_add_u32_e64 v%d, %%5, v%d" % (V_LDS, TMP)]
    T += lds_fetch(SB, V_LDS, 0)
    T += ["s_waitcnt lgkmcnt(0)"]
    g = new_asm(); b = builder(g)
    for src, dst in ((X, AQX), (Y, AQY), (Z, AQZ)):
        b.mov(dst.c0, src.c0.like(1, 0)); b.mov(dst.c1, src.c1.like(1, 0))
    b.mov(AZ3, SA.c0.like(2, 1)); b.mov(ANXZ3, SA.c1.like(3, 1)); b.mov(AXZ, SB.c0.like(2, 1)); b.mov(AYP, SB.c1.like(1, 1))
    T += g.text()
    # subroutine addresses; the bodies sit in front of the loop
    for name, (sreg, _) in SUBS.items():
        T += ["s_getpc_b64 s[%d:%d]" % (sreg, sreg + 1), ".Lml_p%s%%=:" % name,
              "s_add_u32 s%d, s%d, (.Lml_%s%%=-.Lml_p%s%%=)&4294967295" % (sreg, sreg, name, name),
              "s_addc_u32 s%d, s%d, (.Lml_%s%%=-.Lml_p%s%%=)>>32" % (sreg + 1, sreg + 1, name, name)]
    T += ["s_branch .Lml_main%="]
    for name, (_, fn) in SUBS.items():
        a = new_asm(); a.ins = fn()
        T += [".Lml_%s%%=:" % name] + a.text() + ["s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1)]
    T += [".Lml_main%=:", "s_mov_b32 s%d, 0x%x" % (S_XA, X_ABS & 0xffffffff), "s_mov_b32 s%d, 0x%x" % (S_XA + 1, X_ABS >> 32), "s_mov_b32 s%d, 62" % S_I]
    T += [".Lml_loop%=:"] + dbl_ins
    T += ["s_lshr_b64 s[%d:%d], s[%d:%d], s%d" % (S_T, S_T + 1, S_XA, S_XA + 1, S_I), "s_bitcmp1_b32 s%d, 0" % S_T, "s_cbranch_scc1 .Lml_add%="]
    T += [".Lml_next%=:", "s_sub_u32 s%d, s%d, 1" % (S_I, S_I), "s_cmp_ge_i32 s%d, 0" % S_I, "s_cbranch_scc1 .Lml_loop%=", "s_branch .Lml_end%="]
    T += [".Lml_add%=:"] + add_ins + ["s_branch .Lml_next%="]
    T += [".Lml_end%=:", "s_mov_b64 exec, s[%d:%d]" % (S_EXEC, S_EXEC + 1)]
    return T


def clobbers():
    c = ["v%d" % i for i in range(CLOBBER_V)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in range(*CLOBBER_S)] + ["vcc", "scc", "memory"]
    return ", ".join('"%s"' % x for x in c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--quick", action="store_true", help="selftest without the 68-step walk")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    if a.selftest:
        selftest(walk=not a.quick)
        return
    lines = [l if l.startswith(".L") else "\\t" + l for l in kernel_text()]
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_lines_asm.py -- do not edit.\n"
           "// operands: %0 lines base (s pair), %1 row stride in bytes (s), %2 16 * pair index (v), %3 skip flag (v), %4 LDS address of the four hand-over slots (s), %5 of the fifth (s)\n"
           "#define BLS_LINES_ASM_BODY \\\n" + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_LINES_ASM_CLOBBERS " + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
