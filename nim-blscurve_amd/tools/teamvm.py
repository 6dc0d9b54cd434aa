#!/usr/bin/env python3
"""Programs for the lane-team engine of the latency path (csrc/teamvm.hpp): the cofactor clearing of hash-to-G2 and the 68-step Miller
walk of ONE message / pair on a TEAM of 16 lanes (one DPP row), for calls that cannot fill the chip with one lane per item -
fastAggregateVerify, one signature, batches of up to a few thousand sets (reference: bls_sig_min_pubkey.nim:234-258, core :269-297,
blst_abi.nim:383, :455).

The engine, and why it looks like this.  Rounds 1-5 ran these chains as compiled "team" formulas: every lane held
every intermediate of the formula in registers (700 spilled registers), picked its operands with 14-instruction selects and read the
others' results with 14-shuffle gathers: ~5 300 cycles per round for a 1 840-cycle Fp product.  Here the team's values live in LDS as
Fp SLOTS (64 bytes of limbs at a stride of 80) and the formulas are DATA: a program is a list of ROUNDS, a round gives each of the 16 lanes one descriptor

    v    = S[a] * S[b]                      one Montgomery product (392 multiply-adds), operands read from the team's slots by address
    out  = reduce(c0 v + c1 v^1 + c2 v^2 + c3 v^3 + ct S[t])      v^k: the product of lane (l xor k) of the same QUAD (three DPP moves)
    S[dst] = out                            (optionally also a plane of the Miller line store in HBM)

so an Fp2 product is one quad (re = v0 - v1, im = v2 + v3), an Fp2 square three lanes, and a lazily reduced a b - c d (the differences
U2 - U1, S2 - S1, E^2 - 2 D, r (V - X3) - S1 H^3 of the point formulas) costs nothing extra.  Every stored value is partially reduced
(|v| < 0.51 p, canonical limbs), so operands never need a carry and all small multiples (2 Y Z, 3 A, 12 xi C, -8 B^2) are the integer
coefficients c, folded into the reduction's own 64-bit chain.  A LINEAR round skips the product (v = S[a]): sums that must exist as
values (B +- 3 E of the Miller doubling) and copies.  One instruction stream for every round of every formula: ~700 instructions, no
selects, no shuffles, no spills (the kernel holds three Fp values and a descriptor).

This file: the round builder with the engine's bounds asserted while the program is built, the programs (DBL / ADD / PREP / PSI of G2,
the whole cofactor clearing; the Miller doubling and addition steps, the whole 68-step walk), an executor of the same tables on Python
integers (`--selftest`, tests/test_teamvm.py: group law and lines against big-integer formulas, and a table-level simulation that
checks every encoded field), and the emitter of build/teamvm_tables.inc (descriptor words + sequences, included by csrc/kernels.hip).
"""
import argparse
import random
import sys

from asmlib import F2, P, R, RINV, X_ABS, mmul

TEAM = 16
SLOT_BYTES = 80                # 64 bytes of limbs + 16 of padding: slot s starts at bank 20 s mod 32, so eight lanes that read eight different
                               # slots with ds_read_b128 hit eight different bank groups (at a stride of 64 bytes only two: 61 % of the LDS cycles were conflicts)
# sequence entry: round index | flags
F_LINEAR = 1 << 16           # no product: v = S[a]
F_GSTORE = 1 << 17           # lanes with a line plane also write `out` to the line store; bits 20..27: the step
STEP_SHIFT = 20
NO_PLANE = 15


class Slots:
    """named Fp slots of a team's LDS region; an Fp2 value is two consecutive slots"""

    def __init__(self):
        self.n = 0
        self.names = {}

    def fp(self, name):
        self.names[name] = self.n
        self.n += 1
        return self.n - 1

    def fp2(self, name):
        self.names[name] = self.n
        self.n += 2
        return (self.n - 2, self.n - 1)


class Round:
    """16 lane descriptors.  quad(): the next free quad; Quad.mul(a, b) -> product handle; Quad.out(dst, {handle: coef}, t=(ct, slot))."""

    def __init__(self, prog, linear=False, gstore=False):
        self.prog, self.linear, self.gstore = prog, linear, gstore
        self.a = [prog.zero] * TEAM
        self.b = [prog.zero] * TEAM
        self.coef = [[0, 0, 0, 0] for _ in range(TEAM)]          # absolute: coefficient of the product of lane 4 q + j
        self.ct = [0] * TEAM
        self.t = [prog.zero] * TEAM
        self.dst = [prog.trash] * TEAM
        self.plane = [NO_PLANE] * TEAM
        self.nquads = 0

    def quad(self):
        assert self.nquads < TEAM // 4, "round is full (four quads)"
        self.nquads += 1
        return Quad(self, self.nquads - 1)


class Quad:
    def __init__(self, rnd, q):
        self.r, self.q, self.nmul, self.nout = rnd, q, 0, 0

    def mul(self, a, b=None):
        """lane product S[a] * S[b]  (linear round: the value S[a])"""
        assert self.nmul < 4, "quad has four products"
        lane = 4 * self.q + self.nmul
        self.r.a[lane] = a
        if not self.r.linear:
            self.r.b[lane] = b
        else:
            assert b is None
        self.nmul += 1
        return self.nmul - 1

    def out(self, dst, terms, t=None, plane=NO_PLANE):
        """dst <- sum of coef * product[handle] (+ ct * S[slot]); one output per lane"""
        assert self.nout < 4, "quad has four outputs"
        lane = 4 * self.q + self.nout
        for h, c in terms.items():
            assert 0 <= h < 4
            self.r.coef[lane][h] = c
        if t is not None:
            self.r.ct[lane], self.r.t[lane] = t
        total = sum(abs(c) for c in self.r.coef[lane]) + abs(self.r.ct[lane])
        # the engine's reduction: 64-bit chain (no limb overflow), quotient estimated from the top limbs of canonical values - error
        # below total * 2^-17 p; 64 keeps |out| < 0.51 p
        assert total <= 64 and all(-128 <= c <= 127 for c in self.r.coef[lane] + [self.r.ct[lane]]), "coefficient bound"
        self.r.dst[lane] = dst
        self.r.plane[lane] = plane
        self.nout += 1


class Prog:
    def __init__(self, slots):
        self.s = slots
        self.zero = slots.fp("zero")        # holds 0: the operand of idle lanes
        self.trash = slots.fp("trash")      # where idle lanes store
        self.rounds = []                    # distinct rounds
        self.index = {}
        self.seq = []                       # entries: index | flags | step << 20

    def add(self, rnd, step=0):
        key = (rnd.linear, tuple(rnd.a), tuple(rnd.b), tuple(tuple(c) for c in rnd.coef), tuple(rnd.ct), tuple(rnd.t), tuple(rnd.dst), tuple(rnd.plane))
        if key not in self.index:
            self.index[key] = len(self.rounds)
            self.rounds.append(rnd)
        e = self.index[key] | (F_LINEAR if rnd.linear else 0) | (F_GSTORE if rnd.gstore else 0) | (step << STEP_SHIFT)
        self.seq.append(e)

    def round(self, linear=False, gstore=False):
        return Round(self, linear, gstore)


# ---- Fp2-level helpers: one quad each ----------------------------------------------------------------------------------------------------
def q_mul(r, x, y, dst, k=1, t=None, dst2=None):
    """dst <- k * x * y (+ ct * T for an Fp2 value T given as t = (ct, T));  dst2: a second copy of the result"""
    q = r.quad()
    p00, p11, p01, p10 = q.mul(x[0], y[0]), q.mul(x[1], y[1]), q.mul(x[0], y[1]), q.mul(x[1], y[0])
    tr = (t[0], t[1][0]) if t else None
    ti = (t[0], t[1][1]) if t else None
    q.out(dst[0], {p00: k, p11: -k}, tr)
    q.out(dst[1], {p01: k, p10: k}, ti)
    if dst2 is not None:
        q.out(dst2[0], {p00: k, p11: -k}, tr)
        q.out(dst2[1], {p01: k, p10: k}, ti)
    return q


def q_sqr(r, x, dst, k=1):
    q = r.quad()
    a, b, c = q.mul(x[0], x[0]), q.mul(x[1], x[1]), q.mul(x[0], x[1])
    q.out(dst[0], {a: k, b: -k})
    q.out(dst[1], {c: 2 * k})
    return q


def q_mul_conj(r, x, c, dst):
    """dst <- conj(x) * c"""
    q = r.quad()
    p00, p11, p01, p10 = q.mul(x[0], c[0]), q.mul(x[1], c[1]), q.mul(x[0], c[1]), q.mul(x[1], c[0])
    q.out(dst[0], {p00: 1, p11: 1})
    q.out(dst[1], {p01: 1, p10: -1})


def q_diff_re(r, x, y, z, w):
    """products x0 y0, x1 y1, z0 w0, z1 w1: re(x y) = v0 - v1, re(z w) = v2 - v3"""
    q = r.quad()
    return q, (q.mul(x[0], y[0]), q.mul(x[1], y[1]), q.mul(z[0], w[0]), q.mul(z[1], w[1]))


def q_diff_im(r, x, y, z, w):
    """products x0 y1, x1 y0, z0 w1, z1 w0: im(x y) = v0 + v1, im(z w) = v2 + v3"""
    q = r.quad()
    return q, (q.mul(x[0], y[1]), q.mul(x[1], y[0]), q.mul(z[0], w[1]), q.mul(z[1], w[0]))


def copy_round(prog, pairs):
    """linear rounds: dst <- sign * src for (src, dst, sign) Fp slots, sixteen per round"""
    pairs = list(pairs)
    while pairs:
        r = prog.round(linear=True)
        for _ in range(4):
            if not pairs:
                break
            q = r.quad()
            chunk, pairs = pairs[:4], pairs[4:]
            hs = [q.mul(src) for src, _, _ in chunk]
            for h, (_, dst, sg) in zip(hs, chunk):
                q.out(dst, {h: sg})
        prog.add(r)


def copy2(src, dst, sign=1):
    return [(src[0], dst[0], sign), (src[1], dst[1], sign)]


def copy_point(src, dst, neg_y=False):
    out = []
    for i, (s_, d_) in enumerate(zip(src, dst)):
        out += copy2(s_, d_, -1 if (neg_y and i == 1) else 1)
    return out


# ---- G2 Jacobian formulas (y^2 = x^3 + b, a = 0) -------------------------------------------------------------------------------------------
class G2Clear:
    """cofactor clearing H = clear(q0 + q1) (h2c.hpp clear_cofactor_g2_chain; RFC 9380 G.3 / Budroni-Pintore): the accumulator ACC, the
    base BASE with its Z^2, Z^3, temporaries, four parked points."""

    def __init__(self):
        s = self.s = Slots()
        self.p = Prog(s)
        f2 = s.fp2
        self.ACC = (f2("X"), f2("Y"), f2("Z"))
        self.BASE = (f2("BX"), f2("BY"), f2("BZ"))
        self.BZZ, self.BZZZ = f2("BZZ"), f2("BZZZ")
        self.tA, self.tB, self.tW, self.tBB = f2("tA"), f2("tB"), f2("tW"), f2("tBB")
        self.Z1Z1, self.YZ, self.U1, self.S1, self.H, self.RR, self.Z1Z2 = f2("Z1Z1"), f2("YZ"), f2("U1"), f2("S1"), f2("H"), f2("RR"), f2("Z1Z2")
        self.HH, self.R2, self.HHH, self.VX = f2("HH"), f2("R2"), f2("HHH"), f2("VX")
        self.PP, self.PC, self.PT2, self.PU = [(f2(n + "x"), f2(n + "y"), f2(n + "z")) for n in ("PP", "PC", "PT2", "PU")]
        self.CX, self.CY = f2("CX"), f2("CY")          # psi constants (written by the kernel's prologue)
        self.IN0, self.IN1, self.OUT = self.ACC, self.BASE, self.ACC

    # homes <- 2 homes (curve.hpp jac_dbl_lazy): A = X^2, B = Y^2, Z3 = 2 Y Z; X3 = 9 A^2 - 8 X B, W = D - X3 = 12 X B - 9 A^2, B^2;
    # Y3 = 3 A W - 8 B^2
    def dbl(self, src=None):
        X, Y, Z = src or self.ACC
        DX, DY, DZ = self.ACC
        p = self.p
        r = p.round()
        q_sqr(r, X, self.tA)
        q_sqr(r, Y, self.tB)
        q_mul(r, Y, Z, DZ, k=2)
        p.add(r)
        A, B = self.tA, self.tB
        r = p.round()
        q = r.quad()
        a0, a1, x0, x1 = q.mul(A[0], A[0]), q.mul(A[1], A[1]), q.mul(X[0], B[0]), q.mul(X[1], B[1])
        q.out(DX[0], {a0: 9, a1: -9, x0: -8, x1: 8})
        q.out(self.tW[0], {a0: -9, a1: 9, x0: 12, x1: -12})
        q = r.quad()
        aa, xb, bx = q.mul(A[0], A[1]), q.mul(X[0], B[1]), q.mul(X[1], B[0])
        q.out(DX[1], {aa: 18, xb: -8, bx: -8})
        q.out(self.tW[1], {aa: -18, xb: 12, bx: 12})
        q_sqr(r, B, self.tBB)
        p.add(r)
        r = p.round()
        q_mul(r, A, self.tW, DY, k=3, t=(-8, self.tBB))
        p.add(r)

    # BASE's Z^2, Z^3 (curve.hpp jac_precompute); src: where the base comes from (copied into BASE in the same rounds), neg_y: negated
    def prep(self, src=None, neg_y=False):
        p = self.p
        S = src or self.BASE
        r = p.round()
        q_sqr(r, S[2], self.BZZ)
        if src is not None:
            items = copy_point(src, self.BASE, neg_y)
            for k in range(0, len(items), 4):
                q = r.quad()
                for s_, d_, sg in items[k:k + 4]:
                    q.out(d_, {}, t=(sg, s_))
        elif neg_y:
            q = r.quad()
            for c in self.BASE[1]:
                q.out(c, {}, t=(-1, c))
        p.add(r)
        r = p.round()
        q_mul(r, S[2], self.BZZ, self.BZZZ)
        p.add(r)

    # ACC <- ACC + BASE (curve.hpp jac_add_pre).  Exceptional cases (an operand at infinity, P = +-Q) end in Z3 = 0, which every later step
    # keeps: the kernel tests the final Z once and recomputes such a lane with the complete formulas.
    def add(self):
        p = self.p
        X1, Y1, Z1 = self.ACC
        X2, Y2, Z2 = self.BASE
        r = p.round()
        q_sqr(r, Z1, self.Z1Z1)
        q_mul(r, Y2, Z1, self.YZ)
        q_mul(r, X1, self.BZZ, self.U1)
        q_mul(r, Y1, self.BZZZ, self.S1)
        p.add(r)
        r = p.round()
        q_mul(r, X2, self.Z1Z1, self.H, t=(-1, self.U1))                    # H = U2 - U1
        q_mul(r, self.YZ, self.Z1Z1, self.RR, t=(-1, self.S1))              # r = S2 - S1
        q_mul(r, Z1, Z2, self.Z1Z2)
        p.add(r)
        r = p.round()
        q_mul(r, self.Z1Z2, self.H, Z1)                                     # Z3
        q_sqr(r, self.H, self.HH)
        q_sqr(r, self.RR, self.R2)
        p.add(r)
        r = p.round()
        q, (h0, h1, u0, u1) = q_diff_re(r, self.H, self.HH, self.U1, self.HH)
        q.out(X1[0], {h0: -1, h1: 1, u0: -2, u1: 2}, t=(1, self.R2[0]))      # X3 = r^2 - H^3 - 2 V
        q.out(self.HHH[0], {h0: 1, h1: -1})
        q.out(self.VX[0], {h0: 1, h1: -1, u0: 3, u1: -3}, t=(-1, self.R2[0]))  # V - X3 = 3 V + H^3 - r^2
        q, (h0, h1, u0, u1) = q_diff_im(r, self.H, self.HH, self.U1, self.HH)
        q.out(X1[1], {h0: -1, h1: -1, u0: -2, u1: -2}, t=(1, self.R2[1]))
        q.out(self.HHH[1], {h0: 1, h1: 1})
        q.out(self.VX[1], {h0: 1, h1: 1, u0: 3, u1: 3}, t=(-1, self.R2[1]))
        p.add(r)
        r = p.round()
        q, (a, b, c, d) = q_diff_re(r, self.RR, self.VX, self.S1, self.HHH)
        q.out(Y1[0], {a: 1, b: -1, c: -1, d: 1})                             # Y3 = r (V - X3) - S1 H^3
        q, (a, b, c, d) = q_diff_im(r, self.RR, self.VX, self.S1, self.HHH)
        q.out(Y1[1], {a: 1, b: 1, c: -1, d: -1})
        p.add(r)

    # ACC <- psi(src) = (conj(x) cx, conj(y) cy, conj(z))
    def psi(self, src=None):
        S = src or self.ACC
        r = self.p.round()
        q_mul_conj(r, S[0], self.CX, self.ACC[0])
        q_mul_conj(r, S[1], self.CY, self.ACC[1])
        q = r.quad()
        q.out(self.ACC[2][0], {}, t=(1, S[2][0]))
        q.out(self.ACC[2][1], {}, t=(-1, S[2][1]))
        self.p.add(r)

    def chain(self):
        """ACC <- [|x|] ACC"""
        self.prep(src=self.ACC)
        for bit in range(62, -1, -1):
            self.dbl()
            if (X_ABS >> bit) & 1:
                self.add()

    def park(self, dst, neg_y=False):
        copy_round(self.p, copy_point(self.ACC, dst, neg_y))

    def build(self):
        """IN0 = ACC = q0, IN1 = BASE = q1 (written by the prologue) -> OUT = ACC = H"""
        self.prep()
        self.add()                              # P = q0 + q1
        self.park(self.PP)
        self.chain()                            # c = [|x|] P
        self.park(self.PC)
        self.psi(src=self.PP)                   # t2 = psi(P)
        self.park(self.PT2)
        self.dbl(src=self.PP)                   # 2 P
        self.psi()
        self.psi()                              # psi^2(2 P)
        self.prep(src=self.PT2, neg_y=True)
        self.add()                              # u = psi^2(2P) - psi(P)
        self.prep(src=self.PC)
        self.add()                              # u += c       (- [x] P, x < 0)
        self.prep(src=self.PP, neg_y=True)
        self.add()                              # u -= P
        self.park(self.PU)
        copy_round(self.p, copy_point(self.PC, self.ACC, neg_y=True))          # [x] P = -c
        self.prep(src=self.PT2)
        self.add()                              # base = [x] P + psi(P)
        self.chain()                            # [|x|] base
        copy_round(self.p, copy2(self.ACC[1], self.ACC[1], -1))                 # [x] base
        self.prep(src=self.PU)
        self.add()                              # H = u + [x] base
        return self.p


# ---- the Miller walk: T = Q through the 63 doublings + 5 additions of |x|, 68 lines evaluated at P (pairing.hpp) ------------------------------
class Lines:
    def __init__(self):
        s = self.s = Slots()
        self.p = Prog(s)
        f2, f1 = s.fp2, s.fp
        self.PX, self.PY, self.PZ = f1("PX"), f1("PY"), f1("PZ")            # P, Jacobian (prologue)
        self.QJ = (f2("QX"), f2("QY"), f2("QZ"))                            # Q, Jacobian (prologue)
        self.PZ2, self.XZP, self.NXZ3, self.Z3P = f1("PZ2"), f1("XZP"), f1("NXZ3"), f1("Z3P")
        self.QZZ = f2("QZZ")
        self.Q = (f2("QPX"), f2("QPY"), f2("QPZ"))                          # Q homogeneous: (X Z, Y, Z^3)
        self.T = (f2("TX"), f2("TY"), f2("TZ"))
        self.B, self.E, self.Hh, self.XY2 = f2("B"), f2("E"), f2("Hh"), f2("XY2")
        self.BmF, self.BpF, self.BmE, self.XO, self.XK = f2("BmF"), f2("BpF"), f2("BmE"), f2("XO"), f2("XK")
        self.u, self.w, self.Y1Z2, self.X1Z2 = f2("u"), f2("w"), f2("Y1Z2"), f2("X1Z2")
        self.UU, self.WW, self.Z1Z2, self.UX = f2("UU"), f2("WW"), f2("Z1Z2"), f2("UX")
        self.WWW, self.C0, self.C1, self.C2 = f2("WWW"), f2("C0"), f2("C1"), f2("C2")
        self.A, self.RmA, self.VY = f2("A"), f2("RmA"), f2("VY")
        self.L = (f2("L0"), f2("L1"), f2("L2"))                             # the step's line also stays in LDS (tests; the fused consumers)

    def pre(self):
        """g1_precompute (Z^3, X Z, -3 X Z) and g2_to_proj (X Z : Y : Z^3), T = Q"""
        p = self.p
        QX, QY, QZ = self.QJ
        r = p.round()
        q = r.quad()
        zz, xz = q.mul(self.PZ, self.PZ), q.mul(self.PX, self.PZ)
        q.out(self.PZ2, {zz: 1})
        q.out(self.XZP, {xz: 1})
        q.out(self.NXZ3, {xz: -3})
        q_sqr(r, QZ, self.QZZ)
        q_mul(r, QX, QZ, self.Q[0], dst2=self.T[0])
        q = r.quad()
        for d in (self.Q[1], self.T[1]):
            q.out(d[0], {}, t=(1, QY[0]))
            q.out(d[1], {}, t=(1, QY[1]))
        p.add(r)
        r = p.round()
        q = r.quad()
        z3 = q.mul(self.PZ2, self.PZ)
        q.out(self.Z3P, {z3: 1})
        q_mul(r, self.QZZ, QZ, self.Q[2], dst2=self.T[2])
        p.add(r)

    def dbl(self, step):
        """pairing.hpp miller_dbl_step: B = Y^2, C = Z^2, E = 12 xi C, H = 2 Y Z; X3 = 2 X Y (B - 3E), Y3 = (B + 3E)^2 - 12 E^2, Z3 = 4 B H;
        line (B - E) z3p, X^2 nxz3, H yp"""
        p = self.p
        X, Y, Z = self.T
        r = p.round()
        q_sqr(r, Y, self.B)
        q = r.quad()
        a, b, c = q.mul(Z[0], Z[0]), q.mul(Z[1], Z[1]), q.mul(Z[0], Z[1])
        q.out(self.E[0], {a: 12, b: -12, c: -24})                 # 12 xi C, xi = 1 + u: re = 12 (C0 - C1), im = 12 (C0 + C1)
        q.out(self.E[1], {a: 12, b: -12, c: 24})
        q_mul(r, Y, Z, self.Hh, k=2)
        q_mul(r, X, Y, self.XY2, k=2)
        p.add(r)
        r = p.round(linear=True)
        for i in range(2):
            q = r.quad()
            bb, ee, xx = q.mul(self.B[i]), q.mul(self.E[i]), q.mul(X[i])
            q.out(self.BmF[i], {bb: 1, ee: -3})
            q.out(self.BpF[i], {bb: 1, ee: 3})
            q.out(self.BmE[i], {bb: 1, ee: -1})
            q.out(self.XO[i], {xx: 1})
        p.add(r)
        r = p.round()
        q_mul(r, self.XY2, self.BmF, X)
        q = r.quad()
        a, b, c, d = q.mul(self.BpF[0], self.BpF[0]), q.mul(self.BpF[1], self.BpF[1]), q.mul(self.E[0], self.E[0]), q.mul(self.E[1], self.E[1])
        q.out(Y[0], {a: 1, b: -1, c: -12, d: 12})
        q = r.quad()
        a, b, c, d = q.mul(self.BpF[0], self.BpF[1]), q.mul(self.E[0], self.E[1]), q.mul(X[0], self.NXZ3), q.mul(X[1], self.NXZ3)
        q.out(Y[1], {a: 2, b: -24})
        q.out(self.XK[0], {c: 1})
        q.out(self.XK[1], {d: 1})
        q_mul(r, self.B, self.Hh, Z, k=4)
        p.add(r)
        r = p.round(gstore=True)
        q = r.quad()
        a, b, c, d = q.mul(self.BmE[0], self.Z3P), q.mul(self.BmE[1], self.Z3P), q.mul(self.Hh[0], self.PY), q.mul(self.Hh[1], self.PY)
        q.out(self.L[0][0], {a: 1}, plane=0)
        q.out(self.L[0][1], {b: 1}, plane=1)
        q.out(self.L[2][0], {c: 1}, plane=4)
        q.out(self.L[2][1], {d: 1}, plane=5)
        q = r.quad()
        a, b, c = q.mul(self.XO[0], self.XK[0]), q.mul(self.XO[1], self.XK[1]), q.mul(self.XO[0], self.XK[1])
        q.out(self.L[1][0], {a: 1, b: -1}, plane=2)
        q.out(self.L[1][1], {c: 2}, plane=3)
        p.add(r, step)

    def add(self, step):
        """pairing.hpp miller_add_step: u = Y2 Z1 - Y1 Z2, w = X2 Z1 - X1 Z2; A = u^2 Z1Z2 - w^3 - 2 w^2 X1Z2; X3 = w A,
        Y3 = u (w^2 X1Z2 - A) - w^3 Y1Z2, Z3 = w^3 Z1Z2; line (u X2 - w Y2) z3p, -(u Z2) xzp, (w Z2) yp"""
        p = self.p
        X, Y, Z = self.T
        QX, QY, QZ = self.Q
        r = p.round()
        for name, qc, tc in (("u", QY, Y), ("w", QX, X)):
            dst, keep = (self.u, self.Y1Z2) if name == "u" else (self.w, self.X1Z2)
            q, (a, b, c, d) = q_diff_re(r, qc, Z, tc, QZ)
            q.out(dst[0], {a: 1, b: -1, c: -1, d: 1})
            q.out(keep[0], {c: 1, d: -1})
            q, (a, b, c, d) = q_diff_im(r, qc, Z, tc, QZ)
            q.out(dst[1], {a: 1, b: 1, c: -1, d: -1})
            q.out(keep[1], {c: 1, d: 1})
        p.add(r)
        r = p.round()
        q_sqr(r, self.u, self.UU)
        q_sqr(r, self.w, self.WW)
        q_mul(r, Z, QZ, self.Z1Z2)
        q_mul(r, self.u, QX, self.UX)
        p.add(r)
        r = p.round()
        q_mul(r, self.w, self.WW, self.WWW)
        q_mul(r, self.w, QY, self.C0, k=-1, t=(1, self.UX))          # c0 = u X2 - w Y2
        q_mul(r, self.u, QZ, self.C1)
        q_mul(r, self.w, QZ, self.C2)
        p.add(r)
        r = p.round()
        q, (a, b, c, d) = q_diff_re(r, self.UU, self.Z1Z2, self.WW, self.X1Z2)
        q.out(self.A[0], {a: 1, b: -1, c: -2, d: 2}, t=(-1, self.WWW[0]))
        q.out(self.RmA[0], {a: -1, b: 1, c: 3, d: -3}, t=(1, self.WWW[0]))          # R - A, R = w^2 X1Z2
        q, (a, b, c, d) = q_diff_im(r, self.UU, self.Z1Z2, self.WW, self.X1Z2)
        q.out(self.A[1], {a: 1, b: 1, c: -2, d: -2}, t=(-1, self.WWW[1]))
        q.out(self.RmA[1], {a: -1, b: -1, c: 3, d: 3}, t=(1, self.WWW[1]))
        q_mul(r, self.WWW, self.Z1Z2, Z)
        q_mul(r, self.WWW, self.Y1Z2, self.VY)
        p.add(r)
        r = p.round(gstore=True)
        q_mul(r, self.w, self.A, X)
        q_mul(r, self.u, self.RmA, Y, t=(-1, self.VY))
        q = r.quad()
        a, b, c, d = q.mul(self.C0[0], self.Z3P), q.mul(self.C0[1], self.Z3P), q.mul(self.C1[0], self.XZP), q.mul(self.C1[1], self.XZP)
        q.out(self.L[0][0], {a: 1}, plane=0)
        q.out(self.L[0][1], {b: 1}, plane=1)
        q.out(self.L[1][0], {c: -1}, plane=2)
        q.out(self.L[1][1], {d: -1}, plane=3)
        q = r.quad()
        a, b = q.mul(self.C2[0], self.PY), q.mul(self.C2[1], self.PY)
        q.out(self.L[2][0], {a: 1}, plane=4)
        q.out(self.L[2][1], {b: 1}, plane=5)
        p.add(r, step)

    def build(self):
        self.pre()
        step = 0
        for bit in range(62, -1, -1):
            self.dbl(step)
            step += 1
            if (X_ABS >> bit) & 1:
                self.add(step)
                step += 1
        assert step == 68
        return self.p


# ---- execution on Python integers (values are Montgomery images mod p) ------------------------------------------------------------------
def run(prog, S, seq=None, on_gstore=None):
    """executes the rounds of `prog` on the slot list S (integers mod p); every lane reads before any lane writes, as the wave does"""
    for e in (prog.seq if seq is None else seq):
        r = prog.rounds[e & 0xffff]
        linear = bool(e & F_LINEAR)
        v = [S[r.a[l]] if linear else mmul(S[r.a[l]], S[r.b[l]]) for l in range(TEAM)]
        outs = []
        for l in range(TEAM):
            q = l & ~3
            o = sum(r.coef[l][j] * v[q + j] for j in range(4)) + r.ct[l] * S[r.t[l]]
            outs.append(o % P)
        for l in range(TEAM):
            S[r.dst[l]] = outs[l]
            if (e & F_GSTORE) and r.plane[l] != NO_PLANE and on_gstore:
                on_gstore((e >> STEP_SHIFT) & 0xff, r.plane[l], outs[l])
    S[prog.zero] = 0


# ---- the tables: one uint4 per lane and round --------------------------------------------------------------------------------------------
def encode(prog):
    """descriptor words: w0 = a | b << 16 (byte offsets in the team's region), w1 = t | dst << 16, w2 = the four coefficients as signed bytes,
    RELATIVE to the lane (byte k: the product of lane l xor k), w3 = ct (signed byte) | plane << 8"""
    words = []
    for r in prog.rounds:
        for l in range(TEAM):
            j = l & 3
            rel = [r.coef[l][j ^ k] & 0xff for k in range(4)]
            words += [r.a[l] * SLOT_BYTES | (r.b[l] * SLOT_BYTES) << 16, r.t[l] * SLOT_BYTES | (r.dst[l] * SLOT_BYTES) << 16,
                      rel[0] | rel[1] << 8 | rel[2] << 16 | rel[3] << 24, (r.ct[l] & 0xff) | r.plane[l] << 8]
    return words


def run_encoded(words, seq, S):
    """the executor again, from the encoded tables alone (what the kernel sees)"""
    sb = lambda x: x - 256 if x & 0x80 else x
    lines = []
    for e in seq:
        base = (e & 0xffff) * TEAM * 4
        linear = bool(e & F_LINEAR)
        d = [words[base + 4 * l: base + 4 * l + 4] for l in range(TEAM)]
        v = []
        for l in range(TEAM):
            a, b = (d[l][0] & 0xffff) // SLOT_BYTES, (d[l][0] >> 16) // SLOT_BYTES
            v.append(S[a] if linear else mmul(S[a], S[b]))
        outs = []
        for l in range(TEAM):
            o = sum(sb((d[l][2] >> (8 * k)) & 0xff) * v[l ^ k] for k in range(4)) + sb(d[l][3] & 0xff) * S[(d[l][1] & 0xffff) // SLOT_BYTES]
            outs.append(o % P)
        for l in range(TEAM):
            S[(d[l][1] >> 16) // SLOT_BYTES] = outs[l]
            if (e & F_GSTORE) and ((d[l][3] >> 8) & 0xf) != NO_PLANE:
                lines.append(((e >> STEP_SHIFT) & 0xff, (d[l][3] >> 8) & 0xf, outs[l]))
    return lines


# ---- reference formulas (integers mod p, Montgomery images) -----------------------------------------------------------------------------------
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


def ref_dbl(Pt):
    Xr, Yr, Zr = Pt
    A, Bq = Xr.sqr(), Yr.sqr()
    D = (Xr * Bq) * 4
    E = A * 3
    x3 = E.sqr() - D * 2
    return (x3, E * (D - x3) - Bq.sqr() * 8, (Yr * Zr) * 2)


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


def ref_lines(px, py, pz, Q):
    """pairing.hpp miller_lines on Montgomery images: the 68 (l0, l1, l2) triples"""
    z2 = mmul(pz, pz)
    xz = mmul(px, pz)
    z3p, nxz3 = mmul(z2, pz), (-3 * xz) % P
    QX, QY, QZ = Q
    qzz = QZ.sqr()
    q = (QX * QZ, QY, qzz * QZ)
    t = q
    out = []
    xi = lambda v: F2(v.c0 - v.c1, v.c0 + v.c1)
    for bit in range(62, -1, -1):
        X, Y, Z = t
        B, C, X2 = Y.sqr(), Z.sqr(), X.sqr()
        E = xi(C) * 12
        F = E * 3
        H = (Y * Z) * 2
        x3 = ((X * Y) * 2) * (B - F)
        y3 = (B + F).sqr() - E.sqr() * 12
        z3 = (B * H) * 4
        t = (x3, y3, z3)
        out.append(((B - E).mulfp(z3p), X2.mulfp(nxz3), H.mulfp(py)))
        if (X_ABS >> bit) & 1:
            X, Y, Z = t
            Y1Z2, X1Z2, Z1Z2 = Y * q[2], X * q[2], Z * q[2]
            u = q[1] * Z - Y1Z2
            w = q[0] * Z - X1Z2
            uu, ww = u.sqr(), w.sqr()
            www = w * ww
            Rr = ww * X1Z2
            A = uu * Z1Z2 - www - Rr * 2
            t = (w * A, u * (Rr - A) - www * Y1Z2, www * Z1Z2)
            c0, c1, c2 = u * q[0] - w * q[1], u * q[2], w * q[2]
            out.append((c0.mulfp(z3p), -(c1.mulfp(xz)), c2.mulfp(py)))
    return out


def selftest(seed=3, quiet=False):
    rnd = random.Random(seed)
    rf2 = lambda: F2(rnd.randrange(P), rnd.randrange(P))
    put2 = lambda S, sl, v: (S.__setitem__(sl[0], v.c0), S.__setitem__(sl[1], v.c1))
    get2 = lambda S, sl: F2(S[sl[0]], S[sl[1]])
    # ---- cofactor clearing: the formulas are polynomial identities, any triples will do
    g = G2Clear()
    prog = g.build()
    q0, q1 = (rf2(), rf2(), rf2()), (rf2(), rf2(), rf2())
    S = [0] * g.s.n
    for sl, v in zip(g.IN0 + g.IN1, q0 + q1):
        put2(S, sl, v)
    put2(S, g.CX, F2(*PSI_CX)); put2(S, g.CY, F2(*PSI_CY))
    S2 = list(S)
    run(prog, S)
    want = ref_clear(q0, q1)
    for sl, v in zip(g.OUT, want):
        assert get2(S, sl) == v, "cofactor clearing"
    words = encode(prog)
    run_encoded(words, prog.seq, S2)
    for sl, v in zip(g.OUT, want):
        assert get2(S2, sl) == v, "cofactor clearing from the encoded tables"
    nlin = sum(1 for e in prog.seq if e & F_LINEAR)
    if not quiet:
        print("teamvm clear: %d slots, %d distinct rounds, %d in sequence (%d linear)" % (g.s.n, len(prog.rounds), len(prog.seq), nlin))
    # ---- Miller walk
    m = Lines()
    prog = m.build()
    px, py, pz = rnd.randrange(P), rnd.randrange(P), rnd.randrange(P)
    Q = (rf2(), rf2(), rf2())
    S = [0] * m.s.n
    S[m.PX], S[m.PY], S[m.PZ] = px, py, pz
    for sl, v in zip(m.QJ, Q):
        put2(S, sl, v)
    got = {}
    lines = run_encoded(encode(prog), prog.seq, list(S))
    run(prog, S, on_gstore=lambda step, plane, val: got.__setitem__((step, plane), val))
    want = ref_lines(px, py, pz, Q)
    assert len(lines) == 68 * 6 and {(s_, pl): v for s_, pl, v in lines} == got
    for s_, (l0, l1, l2) in enumerate(want):
        for k, f in enumerate((l0, l1, l2)):
            assert got[(s_, 2 * k)] == f.c0 and got[(s_, 2 * k + 1)] == f.c1, ("line", s_, k)
    nlin = sum(1 for e in prog.seq if e & F_LINEAR)
    if not quiet:
        print("teamvm lines: %d slots, %d distinct rounds, %d in sequence (%d linear)" % (m.s.n, len(prog.rounds), len(prog.seq), nlin))
        print("teamvm selftest ok")


def emit(out):
    T = ["// GENERATED by nim-blscurve_amd/tools/teamvm.py -- do not edit.  Programs of the lane-team engine (csrc/teamvm.hpp).\n"
         "// TVM_TABLE: __device__ in the library, static in the host test build.\n"]
    for name, obj in (("CLEAR", G2Clear()), ("LINES", Lines())):
        prog = obj.build()
        words = encode(prog)
        T.append("constexpr uint32_t TVM_%s_SLOTS = %d, TVM_%s_ROUNDS = %d, TVM_%s_NSEQ = %d;\n" % (name, obj.s.n, name, len(prog.rounds), name, len(prog.seq)))
        for k, v in obj.s.names.items():
            T.append("constexpr uint32_t TVM_%s_%s = %d;\n" % (name, k, v))
        T.append("TVM_TABLE __attribute__((aligned(16))) const uint32_t TVM_%s_DESC[%d] = {\n" % (name, len(words)))
        T += ["    " + ", ".join("0x%xu" % w for w in words[i:i + 8]) + ",\n" for i in range(0, len(words), 8)]
        T.append("};\n")
        T.append("TVM_TABLE const uint32_t TVM_%s_SEQ[%d] = {\n" % (name, len(prog.seq) + 2))
        seq = prog.seq + [prog.seq[-1]] * 2           # two entries of padding: the engine reads the sequence two rounds ahead
        T += ["    " + ", ".join("0x%xu" % w for w in seq[i:i + 8]) + ",\n" for i in range(0, len(seq), 8)]
        T.append("};\n")
    txt = "".join(T)
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    if a.selftest:
        selftest()
        return
    emit(a.out)


if __name__ == "__main__":
    main()
