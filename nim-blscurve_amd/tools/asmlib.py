#!/usr/bin/env python3
"""Generator library for the hand-allocated gfx950 loops of the batch path (k_lines, k_hash_clear; k_lineprod's older generator,
gen_lineprod_asm.py, predates it and keeps its own copy of the dot product).

Why assembly, and why this shape: the G2 / Fp12 kernels run one 512-register wave per SIMD, so nothing hides
a wave's own stalls - hipcc's register spills (reloaded right in front of their use), the 28..56 argument moves and two instruction
fetch restarts of every out-of-line multiplier call, callee-entry waits behind stores.  Here every value has a fixed home, the
multipliers are a handful of SUBROUTINES that read their operands from fixed VGPR "slots" (one s_swappc / s_setpc pair per call, no
argument protocol, no waits), and derived operands (sums, differences, multiples) are computed straight into those slots.  The hot code
stays well inside the 64 KB instruction cache.

Three layers:
  Asm      instruction tuples -> text, and a one-lane INTERPRETER that executes the same tuples on Python integers (32-bit registers,
           a 64-bit signed column accumulator with an overflow assertion): a generated loop is tested against big-integer arithmetic on
           the build container, without a GPU (tests/test_asm_loops.py);
  Builder  field operations on 14-limb values (fp.hpp's representation: 28-bit signed limbs, Montgomery R = 2^392) with the SAME
           worst-case bookkeeping as fp.hpp's host tracker (-DBLS_TRACK_BOUNDS): every value carries a value bound vb (multiples of p)
           and a limb bound lb (units of 2^28), every operation asserts its preconditions AT GENERATION TIME;
  Subs     the multiplier bodies: Montgomery dot products of N operand pairs with one reduction (fp.hpp fp_dotn_core), wrapped as
           Fp2 square / Fp2 product / two Fp products by one Fp / plain dot products over the slots.
"""
import random

X_ABS = 0xd201000000010000
XP = -X_ABS
P = (XP - 1) ** 2 * (XP ** 4 - XP ** 2 + 1) // 3 + XP
LB, NL = 28, 14
MASK = (1 << LB) - 1
R = 1 << (LB * NL)
RINV = pow(R, -1, P)
N0 = (-pow(P, -1, 1 << LB)) % (1 << LB)
PL = [(P >> (LB * i)) & MASK for i in range(NL)]
ONE = [((R % P) >> (LB * i)) & MASK for i in range(NL)]
RECIP = 10322735                     # round(2^40 / (p / 2^364)), fp.hpp fp_reduce


def s32(x):
    x &= 0xffffffff
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def limbs_of(x):
    """canonical limbs of 0 <= x < 2^392"""
    return [(x >> (LB * i)) & MASK for i in range(NL)]


def value_of(ls):
    return sum(s32(l) << (LB * i) for i, l in enumerate(ls))


# ---------------------------------------------------------------------------------------------------------------------------------
# instruction level
# ---------------------------------------------------------------------------------------------------------------------------------
class Asm:
    """Instruction tuples.  Register plan of the multiplier: ACC (an even VGPR pair: the column accumulator), M (14 VGPRs: the Montgomery
    quotient digits), SGPRs S_P..S_P+13 (limbs of p), S_N0, S_MASK, S_RECIP."""

    def __init__(self, acc, s_p, s_n0, s_mask, s_recip):
        self.ins = []
        self.ACC, self.S_P, self.S_N0, self.S_MASK, self.S_RECIP = acc, s_p, s_n0, s_mask, s_recip

    def e(self, *t):
        self.ins.append(t)

    # y operand kinds of a multiply-add: ("v", n) VGPR, ("s", n) SGPR, ("c", k) inline constant
    def mad(self, x, y, first=False):
        self.e("mad", x, y, first)

    def text_of(self, t):
        A = self.ACC
        op = t[0]
        if op == "mad":
            _, x, (kind, y), first = t
            ysrc = {"v": "v%d", "s": "s%d", "c": "%d"}[kind] % y
            return "v_mad_i64_i32 v[%d:%d], vcc, v%d, %s, %s" % (A, A + 1, x, ysrc, "0" if first else "v[%d:%d]" % (A, A + 1))
        if op == "mul_lo":
            return "v_mul_lo_u32 v%d, v%d, s%d" % (t[1], t[2], t[3])
        if op == "and_s":
            return "v_and_b32_e64 v%d, s%d, v%d" % (t[1], t[2], t[3])
        if op == "ashr64":
            return "v_ashrrev_i64 v[%d:%d], %d, v[%d:%d]" % (A, A + 1, t[1], A, A + 1)
        if op == "mov":
            return "v_mov_b32_e64 v%d, v%d" % (t[1], t[2])
        if op == "movi":
            return "v_mov_b32_e32 v%d, 0x%x" % (t[1], t[2] & 0xffffffff) if not -16 <= t[2] <= 64 else "v_mov_b32_e64 v%d, %d" % (t[1], t[2])
        if op == "add":
            return "v_add_u32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
        if op == "addi":
            return "v_add_u32_e64 v%d, %d, v%d" % (t[1], t[3], t[2])
        if op == "sub":
            return "v_sub_u32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
        if op == "neg":
            return "v_sub_u32_e64 v%d, 0, v%d" % (t[1], t[2])
        if op == "lshl":
            return "v_lshlrev_b32_e64 v%d, %d, v%d" % (t[1], t[3], t[2])
        if op == "lshl_add":                                   # d = (a << k) + b
            return "v_lshl_add_u32 v%d, v%d, %d, v%d" % (t[1], t[2], t[3], t[4])
        if op == "ashr":
            return "v_ashrrev_i32_e64 v%d, %d, v%d" % (t[1], t[3], t[2])
        if op == "or3":
            return "v_or3_b32 v%d, v%d, v%d, v%d" % (t[1], t[2], t[3], t[4])
        if op == "flag0":                                      # d = 1 where t == 0 (two instructions)
            return "v_cmp_eq_u32_e64 vcc, 0, v%d\nv_cndmask_b32_e64 v%d, v%d, 1, vcc" % (t[2], t[1], t[1])
        if op == "awrite":
            return "v_accvgpr_write_b32 a%d, v%d" % (t[1], t[2])
        if op == "aread":
            return "v_accvgpr_read_b32 v%d, a%d" % (t[1], t[2])
        if op == "call":
            return "s_swappc_b64 s[%d:%d], s[%d:%d]" % (t[2], t[2] + 1, t[3], t[3] + 1)
        if op == "raw":
            return t[1]
        raise ValueError(op)

    def text(self):
        out = []
        for t in self.ins:
            if t[0] != "hook":
                out += self.text_of(t).split("\n")
        return out


class Machine:
    """one lane: v[256], a[256], s{}; subs: name -> instruction list (executed in place of a call)"""

    def __init__(self, asm_proto, subs=None):
        self.v, self.a, self.s = [0] * 256, [0] * 256, {}
        self.ACC = asm_proto.ACC
        for i in range(NL):
            self.s[asm_proto.S_P + i] = PL[i]
        self.s[asm_proto.S_N0], self.s[asm_proto.S_MASK], self.s[asm_proto.S_RECIP] = N0, MASK, RECIP
        self.subs = subs or {}
        self.count = {"valu": 0, "mad": 0, "calls": 0}

    def acc_get(self):
        return s64(self.v[self.ACC] | (self.v[self.ACC + 1] << 32))

    def acc_set(self, x):
        assert -(1 << 63) <= x < (1 << 63), "column accumulator overflow"
        self.v[self.ACC], self.v[self.ACC + 1] = x & 0xffffffff, (x >> 32) & 0xffffffff

    def run(self, ins):
        v, a, s = self.v, self.a, self.s
        for t in ins:
            op = t[0]
            if op == "call":
                self.count["calls"] += 1
                self.run(self.subs[t[1]])
                continue
            if op == "hook":
                t[1](self)
                continue
            if op == "raw":
                continue
            self.count["valu"] += 1
            if op == "mad":
                _, x, (kind, y), first = t
                yv = s32(v[y]) if kind == "v" else (s32(s[y]) if kind == "s" else y)
                self.acc_set((0 if first else self.acc_get()) + s32(v[x]) * yv)
                self.count["mad"] += 1
            elif op == "mul_lo":
                v[t[1]] = (v[t[2]] * s[t[3]]) & 0xffffffff
            elif op == "and_s":
                v[t[1]] = s[t[2]] & v[t[3]]
            elif op == "ashr64":
                self.acc_set(self.acc_get() >> t[1])
            elif op == "mov":
                v[t[1]] = v[t[2]]
            elif op == "movi":
                v[t[1]] = t[2] & 0xffffffff
            elif op == "add":
                v[t[1]] = (v[t[2]] + v[t[3]]) & 0xffffffff
            elif op == "addi":
                v[t[1]] = (v[t[2]] + t[3]) & 0xffffffff
            elif op == "sub":
                v[t[1]] = (v[t[2]] - v[t[3]]) & 0xffffffff
            elif op == "neg":
                v[t[1]] = (-v[t[2]]) & 0xffffffff
            elif op == "lshl":
                v[t[1]] = (v[t[2]] << t[3]) & 0xffffffff
            elif op == "lshl_add":
                v[t[1]] = ((v[t[2]] << t[3]) + v[t[4]]) & 0xffffffff
            elif op == "ashr":
                v[t[1]] = (s32(v[t[2]]) >> t[3]) & 0xffffffff
            elif op == "or3":
                v[t[1]] = v[t[2]] | v[t[3]] | v[t[4]]
            elif op == "flag0":
                self.count["valu"] += 1
                if v[t[2]] == 0:
                    v[t[1]] = 1
            elif op == "awrite":
                a[t[1]] = v[t[2]]
            elif op == "aread":
                v[t[1]] = a[t[2]]
            else:
                raise ValueError(op)


# ---------------------------------------------------------------------------------------------------------------------------------
# field values with worst-case bounds (fp.hpp's tracker, applied while generating)
# ---------------------------------------------------------------------------------------------------------------------------------
class Fp:
    """14 registers (VGPR numbers, or AGPR numbers when agpr=True) + bounds: |value| <= vb p; |limb| < lb 2^28 (+ slack), lb = 0: limbs
    0..12 canonical (non-negative, < 2^28)."""

    def __init__(self, regs, vb=None, lb=None, agpr=False):
        assert len(regs) == NL
        self.r, self.vb, self.lb, self.agpr = list(regs), vb, lb, agpr

    def like(self, vb, lb):
        return Fp(self.r, vb, lb, self.agpr)


def blk(base, agpr=False):
    return Fp([base + i for i in range(NL)], agpr=agpr)


class Fp2:
    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1


def blk2(base, agpr=False):
    return Fp2(blk(base, agpr), blk(base + NL, agpr))


def _u(lb):
    return lb if lb else 1


class Builder:
    """Emits field operations into an Asm.  dst arguments are Fp objects whose registers receive the result; the returned Fp carries the
    result's bounds (same registers).  tmp: one scratch VGPR (carry steps), tmp2: another."""

    def __init__(self, asm, m_regs, tmp, tmp2):
        self.a, self.M, self.tmp, self.tmp2 = asm, m_regs, tmp, tmp2

    # ---- limb-wise
    def _chk(self, vb, lb, what):
        assert lb <= 7 and vb <= 1024, "%s bounds: vb %d lb %d" % (what, vb, lb)

    def add_nc(self, d, x, y):
        for i in range(NL):
            self.a.e("add", d.r[i], x.r[i], y.r[i])
        vb, lb = x.vb + y.vb, _u(x.lb) + _u(y.lb)
        self._chk(vb, lb, "add_nc")
        return d.like(vb, lb)

    def sub_nc(self, d, x, y):
        for i in range(NL):
            self.a.e("sub", d.r[i], x.r[i], y.r[i])
        vb = x.vb + y.vb
        lb = 1 if (x.lb == 0 and y.lb == 0) else _u(x.lb) + _u(y.lb)           # fp_sub_pos: difference of canonical limbs stays within one unit
        self._chk(vb, lb, "sub_nc")
        return d.like(vb, lb)

    def neg(self, d, x):
        for i in range(NL):
            self.a.e("neg", d.r[i], x.r[i])
        return d.like(x.vb, _u(x.lb))

    def mov(self, d, x):
        if d.r == x.r and d.agpr == x.agpr:
            return d.like(x.vb, x.lb)
        for i in range(NL):
            if d.agpr and not x.agpr:
                self.a.e("awrite", d.r[i], x.r[i])
            elif x.agpr and not d.agpr:
                self.a.e("aread", d.r[i], x.r[i])
            else:
                assert not d.agpr
                self.a.e("mov", d.r[i], x.r[i])
        return d.like(x.vb, x.lb)

    def shl(self, d, x, k):                     # x * 2^k
        for i in range(NL):
            self.a.e("lshl", d.r[i], x.r[i], k)
        vb, lb = x.vb << k, _u(x.lb) << k
        self._chk(vb, lb, "shl")
        return d.like(vb, lb)

    def shl_add(self, d, x, k, y):              # x * 2^k + y
        for i in range(NL):
            self.a.e("lshl_add", d.r[i], x.r[i], k, y.r[i])
        vb, lb = (x.vb << k) + y.vb, (_u(x.lb) << k) + _u(y.lb)
        self._chk(vb, lb, "shl_add")
        return d.like(vb, lb)

    def mul3(self, d, x):
        return self.shl_add(d, x, 1, x)

    def carry(self, d, x):
        """fp_carry_step: limbs 0..12 into [-4, 2^28 + 4), limb 13 absorbs the rest.  d may be x (top-down, one temporary)."""
        assert x.lb <= 7, "carry limb bound"
        t = self.tmp
        for i in range(NL - 1, 0, -1):
            self.a.e("ashr", t, x.r[i - 1], 28)
            if i == NL - 1:
                self.a.e("add", d.r[i], x.r[i], t)
            else:
                self.a.e("and_s", d.r[i], self.a.S_MASK, x.r[i])
                self.a.e("add", d.r[i], d.r[i], t)
        self.a.e("and_s", d.r[0], self.a.S_MASK, x.r[0])
        return d.like(x.vb, 1)

    def reduce(self, d, x, scale=1):
        """fp_reduce of scale * x (scale: a small integer, an inline constant -16 .. 64, folded into the chain as a multiplier - no
        limb-wise multiple, no carry step in front, and a negative one gives the negated value for free): subtracts round(scale x / p) p, estimated from the top limbs; |result| < 0.51 p, canonical limbs.
        d may be x."""
        assert -16 <= scale <= 64 and scale != 0 and x.vb * abs(scale) <= 1024 and _u(x.lb) * abs(scale) <= 56, "reduce bound"
        a, t, t2 = self.a, self.tmp, self.tmp2
        a.e("ashr", t, x.r[NL - 2], 28)
        a.e("add", t, x.r[NL - 1], t)                          # top = a13 + (a12 >> 28)
        if scale != 1:
            a.mad(t, ("c", scale), True)                       # (scale top) as a 64-bit value, then times RECIP: two steps, scale * RECIP may not fit the inline range
            a.e("mov", t, a.ACC)
        a.mad(t, ("s", a.S_RECIP), True)
        a.e("ashr64", 39)
        a.e("addi", t, a.ACC, 1)
        a.e("ashr", t, t, 1)                                   # q = (top RECIP + 2^39) >> 40
        a.e("neg", t, t)                                       # -q
        for i in range(NL):
            a.mad(x.r[i], ("c", scale), i == 0)
            a.mad(t, ("s", a.S_P + i))
            if i < NL - 1:
                a.e("and_s", d.r[i], a.S_MASK, a.ACC)
                a.e("ashr64", 28)
            else:
                a.e("mov", d.r[i], a.ACC)
        return d.like(1, 0)

    def zero_test(self, dst, parts, scratch):
        """dst (a VGPR) <- the OR of all limbs of the partially reduced values `parts`: zero exactly where every one of them is 0 mod p"""
        regs = []
        for x, sc in zip(parts, scratch):
            regs += self.reduce(sc, x).r
        self.a.e("or3", dst, regs[0], regs[1], regs[2])
        rest = regs[3:]
        while rest:
            if len(rest) >= 2:
                self.a.e("or3", dst, dst, rest[0], rest[1])
                rest = rest[2:]
            else:
                self.a.e("or3", dst, dst, rest[0], rest[0])
                rest = rest[1:]

    def flag_if_zero(self, flag, parts, scratch):
        """flag (a VGPR) <- 1 where every one of the values `parts` (Fp objects) is 0 mod p: each is partially reduced (|r| < 0.51 p, so
        0 mod p means all limbs zero) into `scratch` (one Fp block per part) and the limbs are OR-ed together."""
        regs = []
        for x, sc in zip(parts, scratch):
            regs += self.reduce(sc, x).r
        t = self.tmp2
        self.a.e("or3", t, regs[0], regs[1], regs[2])
        rest = regs[3:]
        while rest:
            if len(rest) >= 2:
                self.a.e("or3", t, t, rest[0], rest[1])
                rest = rest[2:]
            else:
                self.a.e("or3", t, t, rest[0], rest[0])
                rest = rest[1:]
        self.a.e("flag0", flag, t)

    # ---- the multiplier: Montgomery dot product of operand pairs -> dst (may be M or any operand block, see gen_lineprod_asm.py)
    def dot_body(self, pairs, dst):
        a, M = self.a, self.M
        first = True
        for kk in range(2 * NL - 1):
            lo, hi = max(0, kk - NL + 1), min(kk, NL - 1)
            for x, y in pairs:
                for i in range(lo, hi + 1):
                    a.mad(x.r[i], ("v", y.r[kk - i]), first)
                    first = False
            if kk < NL:
                for i in range(kk):
                    a.mad(M[i], ("s", a.S_P + kk - i))
                a.e("mul_lo", M[kk], a.ACC, a.S_N0)
                a.e("and_s", M[kk], a.S_MASK, M[kk])
                a.mad(M[kk], ("s", a.S_P))
            else:
                for i in range(kk - NL + 1, NL):
                    a.mad(M[i], ("s", a.S_P + kk - i))
                a.e("and_s", dst.r[kk - NL], a.S_MASK, a.ACC)
            a.e("ashr64", 28)
        a.e("mov", dst.r[NL - 1], a.ACC)

    def sqr_body(self, x, x2, dst):
        """Montgomery square (fp.hpp fp_sqr_core): the 91 off-diagonal products once against the doubled operand x2 = 2 x (the caller
        provides the 14 registers), so the operand half costs 105 multiply-adds instead of 196.  dst may be x, x2 or M."""
        a, M = self.a, self.M
        for i in range(NL):
            a.e("lshl", x2.r[i], x.r[i], 1)
        first = True
        for kk in range(2 * NL - 1):
            lo = 0 if kk < NL else kk - NL + 1
            i = lo
            while 2 * i < kk:
                a.mad(x2.r[i], ("v", x.r[kk - i]), first)
                first = False
                i += 1
            if kk % 2 == 0:
                a.mad(x.r[kk // 2], ("v", x.r[kk // 2]), first)
                first = False
            if kk < NL:
                for i in range(kk):
                    a.mad(M[i], ("s", a.S_P + kk - i))
                a.e("mul_lo", M[kk], a.ACC, a.S_N0)
                a.e("and_s", M[kk], a.S_MASK, M[kk])
                a.mad(M[kk], ("s", a.S_P))
            else:
                for i in range(kk - NL + 1, NL):
                    a.mad(M[i], ("s", a.S_P + kk - i))
                a.e("and_s", dst.r[kk - NL], a.S_MASK, a.ACC)
            a.e("ashr64", 28)
        a.e("mov", dst.r[NL - 1], a.ACC)

    def sqr(self, x, x2, dst):
        assert x.vb * x.vb <= 2048 and _u(x.lb) <= 2, "sqr operand bounds"
        self.sqr_body(x, x2, dst)
        return dst.like(2, 0)

    def dot(self, pairs, dst):
        """checked form: fp_dotn's preconditions"""
        vsum = sum(x.vb * y.vb for x, y in pairs)
        lsum = sum(_u(x.lb) * _u(y.lb) for x, y in pairs)
        assert vsum <= 2048, "dot value bounds %d" % vsum
        assert lsum <= 8, "dot limb-unit bounds %d" % lsum
        self.dot_body(pairs, dst)
        return dst.like(2, 0)


def dot_bounds_ok(pairs):
    return sum(x.vb * y.vb for x, y in pairs) <= 2048 and sum(_u(x.lb) * _u(y.lb) for x, y in pairs) <= 8


# ---------------------------------------------------------------------------------------------------------------------------------
# reference arithmetic for the self-tests: values are Montgomery images, a product is a b / R
# ---------------------------------------------------------------------------------------------------------------------------------
def mmul(a, b):
    return a * b * RINV % P


class F2:
    """Fp2 element as a pair of integers mod p (Montgomery images)"""

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0 % P, c1 % P

    def __add__(self, o):
        return F2(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return F2(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return F2(-self.c0, -self.c1)

    def __mul__(self, o):
        if isinstance(o, int):
            return F2(self.c0 * o, self.c1 * o)
        return F2(mmul(self.c0, o.c0) - mmul(self.c1, o.c1), mmul(self.c0, o.c1) + mmul(self.c1, o.c0))

    def sqr(self):
        return self * self

    def mulfp(self, k):                      # times an Fp element (Montgomery image)
        return F2(mmul(self.c0, k), mmul(self.c1, k))

    def xi(self):
        return F2(self.c0 - self.c1, self.c0 + self.c1)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def __repr__(self):
        return "F2(%x, %x)" % (self.c0, self.c1)


def rand_fp(rnd, signed_limbs=False):
    """a value as the kernels store it: canonical limbs of a residue, or (signed_limbs) a limb-wise negated one"""
    x = rnd.randrange(P)
    if signed_limbs and rnd.random() < 0.5:
        return [(-l) & 0xffffffff for l in limbs_of((P - x) % P)], x
    return limbs_of(x), x


def put(mach, fp, limbs):
    for r, l in zip(fp.r, limbs):
        (mach.a if fp.agpr else mach.v)[r] = l & 0xffffffff


def get(mach, fp):
    return value_of([(mach.a if fp.agpr else mach.v)[r] for r in fp.r])


def check_limbs(mach, fp, lb):
    """the stored limbs respect the declared limb bound"""
    lim = (_u(lb) << LB) + (1 << 20)
    ls = [s32((mach.a if fp.agpr else mach.v)[r]) for r in fp.r]
    assert all(-lim < l < lim for l in ls[:NL - 1]), ("limb magnitude exceeds declared bound", lb, [hex(l) for l in ls])
    if lb == 0:
        assert all(0 <= l < (1 << LB) for l in ls[:NL - 1]), "limbs not canonical"
