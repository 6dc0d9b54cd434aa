#!/usr/bin/env python3
"""Generates the body of k_pip_bucket<fp> (csrc/kernels.hip), the bucket accumulation of the G1 Pippenger MSM
(blst_p1s_mult_pippenger, blst_abi.nim:336-340; benchmarks/bls12381_msm_g1.nim), as ONE hand-allocated gfx950 inline-asm statement:
a lane owns a bucket, walks the bucket's sorted (point index, sign) list and adds the points into an accumulator in extended Jacobian
coordinates (curve.hpp xyzz_add_aff: X, Y, ZZ, ZZZ; 6 products, 2 squares and one lazily reduced a b - c d per mixed addition), then
stores the bucket sum in Jacobian form (X ZZ, Y ZZZ, ZZ).

Why assembly here (profiles/r04_pmc_summary_msm.json: the compiled kernel is at 0.48 of the multiply-add peak): 66 % of its instructions
are multiply-adds (out-of-line multipliers: argument moves, ten calls per addition), 12 % of its wave cycles wait for the dependent
index -> point gather - a callee's entry wait drains every outstanding load, so the next point cannot be prefetched across the calls.
Here G1 arithmetic needs no subroutines: the nine multiplier bodies of an addition are expanded in place on fixed registers (36 KB of
loop code, no operand copies at all), the NEXT point's 128 bytes are gathered while the current addition runs (its index was loaded an
iteration earlier still), and the whole thing fits 230 VGPRs, i.e. two waves per SIMD as before.
Exceptional cases of the complete formula (the addend equals the accumulator: a doubling; equals its negative: the sum is infinity; an
input point at infinity) are not handled in the loop: ZZ3 = 0 after an addition or an infinity flag on a gathered point raises a per-lane
FLAG and the kernel recomputes that bucket with the compiled complete formulas (a few hundred of 524 288 buckets in the benchmark's
input, which repeats its 2 048 points).

`--selftest`: asmlib's interpreter against big-integer arithmetic (first point, additions with both signs, the final conversion, the flag).
"""
import argparse
import random
import sys

from asmlib import Asm, Builder, Fp, Machine, MASK, N0, NL, ONE, P, PL, RECIP, blk, check_limbs, get, limbs_of, mmul, put

# ---- register plan (VGPRs only: two waves per SIMD) --------------------------------------------------------------------------------------
AX, AY, AZZ, AZZZ = blk(0), blk(14), blk(28), blk(42)
QX, QY = blk(56), blk(70)
NX0, NY0 = 84, 100                               # the gathered record of the NEXT point: x (14 limbs, infinity flag, pad), y (14 limbs, 2 pad words)
NX, NY = blk(NX0), blk(NY0)
M_REGS = [116 + i for i in range(NL)]
T = [blk(130 + 14 * i) for i in range(6)]
ACC, TMP, TMP2 = 214, 216, 217
V_E, V_EN, V_PTR, V_PA, V_CNT, V_FLAG, V_S, V_OFF, V_ZP, V_AINF = 218, 219, 220, 222, 224, 225, 226, 227, 228, 229
CLOBBER_V = 230
S_P, S_N0, S_MASK, S_RECIP = 36, 50, 51, 52
S_PTS, S_J, S_FOUR, S_OUT, S_OSTR, S_T, S_EXEC, S_GP, S_LIVE, S_ZP, S_AI, S_SAVE = 54, 56, 58, 60, 62, 64, 66, 68, 70, 72, 74, 76
CLOBBER_S = (36, 78)


def new_asm():
    return Asm(ACC, S_P, S_N0, S_MASK, S_RECIP)


def builder(a):
    return Builder(a, M_REGS, TMP, TMP2)


def gen_add():
    """acc <- acc + q (curve.hpp xyzz_add_aff_flag without its exceptional branches); flag if ZZ3 == 0"""
    a = new_asm(); b = builder(a)
    ax, ay, azz, azzz = AX.like(2, 0), AY.like(2, 1), AZZ.like(2, 0), AZZZ.like(2, 0)      # the widest accumulator: a first point taken as is (y possibly negated)
    qx, qy = QX.like(2, 0), QY.like(2, 1)                   # the internal point records are products (|v| < 2 p); y possibly negated limb-wise
    T0, T1, T2, T3, T4, T5 = T
    u2 = b.dot([(qx, azz)], T0)
    s2 = b.dot([(qy, azzz)], T1)
    p_ = b.sub_nc(T0, u2, ax)                               # P = U2 - X1
    b.zero_test(V_ZP, [p_], [T4])                           # zero where the addend's x equals the accumulator's: P == +-Q, the rare path's business
    r_ = b.sub_nc(T1, s2, ay)                               # R = S2 - Y1
    pp = b.sqr(p_, T5, T2)
    ppp = b.dot([(p_, pp)], T3)
    q_ = b.dot([(ax, pp)], T0)                              # Q = X1 PP (P is dead)
    zz3 = b.dot([(azz, pp)], AZZ)
    zzz3 = b.dot([(azzz, ppp)], AZZZ)
    r2 = b.sqr(r_, T5, T4)
    t = b.sub_nc(T4, r2, ppp)
    t = b.sub_nc(T4, t, b.shl(T5, q_, 1))
    x3 = b.reduce(AX, t)
    qx3 = b.sub_nc(T0, q_, x3)
    ny1 = b.neg(T5, ay)
    y3 = b.dot([(r_, qx3), (ny1, ppp)], AY)
    return a.ins, (x3, y3, zz3, zzz3)


def gen_rtest():
    """rare path: V_ZP <- zero where R == 0 (the addend IS the accumulator's point: a doubling)"""
    a = new_asm(); b = builder(a)
    b.zero_test(V_ZP, [T[1].like(4, 2)], [T[4]])
    return a.ins


def gen_dbl():
    """rare path: acc <- 2 q for an affine q (curve.hpp xyzz_dbl_aff; q is not the point at infinity)"""
    a = new_asm(); b = builder(a)
    qx, qy = QX.like(2, 0), QY.like(2, 1)
    T0, T1, T2, T3, T4, T5 = T
    u = b.carry(T0, b.shl(T0, qy, 1))                       # U = 2 y
    v = b.sqr(u, T5, T1)                                    # V = U^2
    w = b.dot([(u, v)], T2)                                 # W = U V
    s_ = b.dot([(qx, v)], T3)                               # S = x V
    xx = b.sqr(qx, T5, T4)
    m = b.carry(T4, b.mul3(T4, xx))                         # M = 3 x^2
    m2 = b.sqr(m, T5, T0)
    x3 = b.reduce(AX, b.sub_nc(T0, m2, b.shl(T5, s_, 1)))
    sx = b.sub_nc(T3, s_, x3)
    nw = b.neg(T5, w)
    y3 = b.dot([(m, sx), (nw, qy)], AY)
    b.mov(AZZ, v); b.mov(AZZZ, w)
    return a.ins


def gen_take():
    """q <- the gathered record (x as is, y negated where the sign bit of the index word is set); an infinity flag on the record raises the lane's flag"""
    a = new_asm(); b = builder(a)
    a.e("ashr", V_S, V_E, 31)                               # -1 where the point enters negated
    for i in range(NL):
        a.e("mov", QX.r[i], NX.r[i])
    for i in range(NL):                                     # (y ^ s) - s
        a.e("raw_xor", QY.r[i], NY.r[i], V_S)
        a.e("sub", QY.r[i], QY.r[i], V_S)
    a.e("raw_or", V_FLAG, V_FLAG, NX0 + 14)
    return a.ins


def gen_first():
    """acc <- (x, +-y, 1, 1) from q"""
    a = new_asm(); b = builder(a)
    b.mov(AX, QX.like(2, 0)); b.mov(AY, QY.like(2, 1))
    for dst in (AZZ, AZZZ):
        for r, l in zip(dst.r, ONE):
            a.e("movi", r, l)
    return a.ins


def gen_final():
    """(X ZZ, Y ZZZ) -> T0, T1; Z = ZZ stays in AZZ"""
    a = new_asm(); b = builder(a)
    b.dot([(AX.like(2, 0), AZZ.like(2, 0))], T[0])
    b.dot([(AY.like(2, 1), AZZZ.like(2, 0))], T[1])
    return a.ins


class MsmAsm(Asm):
    def text_of(self, t):
        if t[0] == "raw_xor":
            return "v_xor_b32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
        if t[0] == "raw_or":
            return "v_or_b32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
        return Asm.text_of(self, t)


class MsmMachine(Machine):
    def run(self, ins):
        rest = []
        for t in ins:
            if t[0] in ("raw_xor", "raw_or"):
                Machine.run(self, rest); rest = []
                self.count["valu"] += 1
                self.v[t[1]] = (self.v[t[2]] ^ self.v[t[3]]) if t[0] == "raw_xor" else (self.v[t[2]] | self.v[t[3]])
            else:
                rest.append(t)
        Machine.run(self, rest)


def text_of_list(ins):
    a = MsmAsm(ACC, S_P, S_N0, S_MASK, S_RECIP); a.ins = ins
    return a.text()


# ---- reference: the same formulas on big integers (Montgomery images) -----------------------------------------------------------------------
def ref_add(acc, q):
    x1, y1, zz, zzz = acc
    x2, y2 = q
    u2, s2 = mmul(x2, zz), mmul(y2, zzz)
    p_, r_ = (u2 - x1) % P, (s2 - y1) % P
    pp = mmul(p_, p_)
    ppp, q_ = mmul(p_, pp), mmul(x1, pp)
    x3 = (mmul(r_, r_) - ppp - 2 * q_) % P
    y3 = (mmul(r_, (q_ - x3) % P) - mmul(y1, ppp)) % P
    return (x3, y3, mmul(zz, pp), mmul(zzz, ppp))


def ref_dbl_aff(q):
    x, y = q
    u = 2 * y % P
    v = mmul(u, u)
    w, s_, xx = mmul(u, v), mmul(x, v), mmul(x, x)
    m = 3 * xx % P
    x3 = (mmul(m, m) - 2 * s_) % P
    return (x3, (mmul(m, (s_ - x3) % P) - mmul(w, y)) % P, v, w)


def lane_step(mach, parts):
    """what ONE lane of the wave does for one list entry, following the kernel text's control flow: the addition, then - only where
    P == 0 or the accumulator was the point at infinity - the rare path"""
    take, add_ins, first, rtest, dbl = parts
    mach.run(take)
    mach.run(add_ins)
    if mach.v[V_AINF]:
        mach.run(first)
        mach.v[V_AINF] = 0
    elif mach.v[V_ZP] == 0:
        mach.run(rtest)
        if mach.v[V_ZP] == 0:
            mach.run(dbl)
        else:
            mach.v[V_AINF] = 1


def selftest(seed=9):
    rnd = random.Random(seed)
    add_ins, outs = gen_add()
    take, first, final, rtest, dbl = gen_take(), gen_first(), gen_final(), gen_rtest(), gen_dbl()
    parts = (take, add_ins, first, rtest, dbl)
    for o_, bd in zip(outs, ((1, 0), (2, 0), (2, 0), (2, 0))):       # the accumulator's bounds are a fixed point of the addition
        assert (o_.vb, o_.lb) == bd
    mach = MsmMachine(new_asm())
    R1 = (1 << 392) % P

    def gather(x, y, neg, inf=0):
        put(mach, NX, limbs_of(x)); put(mach, NY, limbs_of(y))
        mach.v[NX0 + 14] = inf
        mach.v[V_E] = (0x80000000 if neg else 0) | rnd.randrange(1 << 20)

    def same_point(acc, ref):
        """extended Jacobian triples as points: x ZZ' == x' ZZ, y ZZZ' == y' ZZZ"""
        return mmul(acc[0], ref[2]) == mmul(ref[0], acc[2]) and mmul(acc[1], ref[3]) == mmul(ref[1], acc[3])

    def regs_acc():
        return tuple(get(mach, r) % P for r in (AX, AY, AZZ, AZZZ))

    pts = [(rnd.randrange(P), rnd.randrange(P), rnd.random() < 0.5) for _ in range(12)]
    mach.v[V_AINF] = 1                                       # an empty accumulator: the first entry takes the rare path's "was infinity" branch
    x, y, ng = pts[0]
    gather(x, y, ng); lane_step(mach, parts)
    acc = (x, (-y if ng else y) % P, R1, R1)
    assert regs_acc() == acc and mach.v[V_AINF] == 0
    n0 = dict(mach.count)
    for x, y, ng in pts[1:]:
        gather(x, y, ng); lane_step(mach, parts)
        acc = ref_add(acc, (x, (-y if ng else y) % P))
        for reg, ref, o_ in zip((AX, AY, AZZ, AZZZ), acc, outs):
            assert get(mach, reg) % P == ref, "accumulator mismatch"
            check_limbs(mach, reg, o_.lb)
            assert abs(get(mach, reg)) <= o_.vb * P
    n1 = dict(mach.count)
    assert mach.v[V_FLAG] == 0 and mach.v[V_AINF] == 0
    mach.run(final)
    assert get(mach, T[0]) % P == mmul(acc[0], acc[2]) and get(mach, T[1]) % P == mmul(acc[1], acc[3])
    # ---- the exceptional cases, in the loop itself: q, q (a doubling), then -2q ... wait: acc = 2q, add -q twice -> q, then infinity, then a fresh point
    x, y, ng = pts[0]
    q = (x, (-y if ng else y) % P)
    mach.v[V_AINF] = 1
    gather(x, y, ng); lane_step(mach, parts)                 # acc = q
    gather(x, y, ng); lane_step(mach, parts)                 # acc = 2 q through the doubling branch
    assert mach.v[V_AINF] == 0 and regs_acc() == ref_dbl_aff(q)
    for reg, bd in zip((AX, AY, AZZ, AZZZ), ((2, 0), (2, 1), (2, 0), (2, 0))):
        check_limbs(mach, reg, bd[1])
    gather(x, y, not ng); lane_step(mach, parts)             # 2 q - q = q (an ordinary addition)
    assert same_point(regs_acc(), (q[0], q[1], R1, R1))
    gather(x, y, not ng); lane_step(mach, parts)             # q - q = infinity
    assert mach.v[V_AINF] == 1
    x2, y2, ng2 = pts[3]
    gather(x2, y2, ng2); lane_step(mach, parts)              # infinity + r = r
    assert mach.v[V_AINF] == 0 and regs_acc() == (x2, (-y2 if ng2 else y2) % P, R1, R1)
    assert mach.v[V_FLAG] == 0
    gather(x, y, ng, inf=1); mach.run(take)
    assert mach.v[V_FLAG] == 1, "an input point at infinity raises the lane's flag (recomputed with the compiled formulas)"
    per = {k: (n1[k] - n0[k]) // (len(pts) - 1) for k in n0}
    print("gen_msm_asm selftest ok: %d VALU instructions per addition (%d multiply-adds, %.1f %%); rare path: doubling %d instructions"
          % (per["valu"], per["mad"], 100.0 * per["mad"] / per["valu"], len([t for t in dbl if t[0] not in ("raw", "hook")])))


# ---- text ---------------------------------------------------------------------------------------------------------------------------
def gather_point_text(idx_reg):
    """point record of index word idx_reg (bit 31 = sign) -> N registers: 8 x 16 bytes from pts + 128 * index"""
    t = ["v_and_b32_e32 v%d, 0x7fffffff, v%d" % (V_PA, idx_reg), "v_mov_b32_e64 v%d, 0" % (V_PA + 1),
         "v_lshlrev_b64 v[%d:%d], 7, v[%d:%d]" % (V_PA, V_PA + 1, V_PA, V_PA + 1),
         "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (V_PA, V_PA + 1, V_PA, V_PA + 1, S_PTS, S_PTS + 1)]
    for q in range(4):
        t.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off offset:%d" % (NX0 + 4 * q, NX0 + 4 * q + 3, V_PA, V_PA + 1, 16 * q))
    for q in range(4):
        t.append("global_load_dwordx4 v[%d:%d], v[%d:%d], off offset:%d" % (NY0 + 4 * q, NY0 + 4 * q + 3, V_PA, V_PA + 1, 64 + 16 * q))
    return t


def kernel_text_pipelined():
    """The loop, software-pipelined by one record: while addition j runs, the 128 bytes of point j + 1 are in flight (its index word was loaded
    during addition j - 1) and the index word of point j + 2 is requested."""
    add_ins, _ = gen_add()
    T_ = []
    T_ += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T_ += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK), "s_mov_b32 s%d, 0x%x" % (S_RECIP, RECIP)]
    T_ += ["s_mov_b64 s[%d:%d], %%1" % (S_PTS, S_PTS + 1), "s_mov_b32 s%d, 4" % S_FOUR, "s_mov_b32 s%d, 0" % (S_FOUR + 1),
           "s_mov_b64 s[%d:%d], %%5" % (S_OUT, S_OUT + 1), "s_mov_b32 s%d, %%6" % S_OSTR, "s_mov_b64 s[%d:%d], exec" % (S_EXEC, S_EXEC + 1)]
    T_ += ["v_mov_b32_e64 v%d, %%2" % V_PTR, "v_mov_b32_e64 v%d, %%3" % (V_PTR + 1), "v_mov_b32_e64 v%d, %%4" % V_CNT, "v_mov_b32_e64 v%d, %%7" % V_OFF,
           "v_mov_b32_e64 v%d, 0" % V_FLAG, "v_mov_b32_e64 v%d, 1" % V_AINF]       # an empty accumulator is the point at infinity
    adv = "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (V_PTR, V_PTR + 1, V_PTR, V_PTR + 1, S_FOUR, S_FOUR + 1)

    def masked(cond_k, body):
        """body under exec &= (k < cnt), k = S_J + cond_k; exec restored"""
        pre = ["s_add_u32 s%d, s%d, %d" % (S_LIVE, S_J, cond_k)] if cond_k else ["s_mov_b32 s%d, s%d" % (S_LIVE, S_J)]
        return pre + ["v_cmp_lt_u32_e64 vcc, s%d, v%d" % (S_LIVE, V_CNT), "s_and_saveexec_b64 s[%d:%d], vcc" % (S_T, S_T + 1)] + body + \
               ["s_mov_b64 exec, s[%d:%d]" % (S_T, S_T + 1)]
    # ---- entry 0: its index, its record; index 1
    T_ += ["s_mov_b32 s%d, 0" % S_J]
    T_ += masked(0, ["global_load_dword v%d, v[%d:%d], off" % (V_E, V_PTR, V_PTR + 1)])
    T_ += masked(1, ["global_load_dword v%d, v[%d:%d], off offset:4" % (V_EN, V_PTR, V_PTR + 1)])
    T_ += [adv, adv, "s_waitcnt vmcnt(0)"]
    T_ += masked(0, gather_point_text(V_E) + ["s_waitcnt vmcnt(0)"] + text_of_list(gen_take()) + text_of_list(gen_first()) + ["v_mov_b32_e64 v%d, 0" % V_AINF])
    # record 1 and index 2 go out; then the loop: j = 1 ..
    T_ += ["s_mov_b32 s%d, 1" % S_J]
    T_ += masked(0, gather_point_text(V_EN) + ["v_mov_b32_e64 v%d, v%d" % (V_E, V_EN)])
    T_ += masked(1, ["global_load_dword v%d, v[%d:%d], off" % (V_EN, V_PTR, V_PTR + 1)]) + [adv]
    T_ += [".Lmb_loop%=:",
           "v_cmp_lt_u32_e64 vcc, s%d, v%d" % (S_J, V_CNT), "s_and_b64 exec, exec, vcc", "s_cbranch_execz .Lmb_store%=",
           "s_waitcnt vmcnt(0)"]                                 # record j and index word j + 1 have arrived
    T_ += text_of_list(gen_take())                               # q <- record j (sign from V_E)
    T_ += masked(1, gather_point_text(V_EN) + ["v_mov_b32_e64 v%d, v%d" % (V_E, V_EN)])          # record j + 1 goes out; its index word becomes the current one
    T_ += masked(2, ["global_load_dword v%d, v[%d:%d], off" % (V_EN, V_PTR, V_PTR + 1)]) + [adv]   # index word j + 2
    T_ += text_of_list(add_ins)
    # exceptional lanes: P == 0 (the addend is +- the accumulator's point) or the accumulator was the point at infinity -> the rare path, out of line
    T_ += ["v_cmp_eq_u32_e64 s[%d:%d], 0, v%d" % (S_ZP, S_ZP + 1, V_ZP), "v_cmp_ne_u32_e64 s[%d:%d], 0, v%d" % (S_AI, S_AI + 1, V_AINF),
           "s_or_b64 vcc, s[%d:%d], s[%d:%d]" % (S_ZP, S_ZP + 1, S_AI, S_AI + 1), "s_cbranch_vccnz .Lmb_rare%=", ".Lmb_back%=:"]
    T_ += ["s_add_u32 s%d, s%d, 1" % (S_J, S_J), "s_branch .Lmb_loop%="]
    # ---- the rare path (exec = the active lanes of this iteration on entry and on exit)
    T_ += [".Lmb_rare%=:", "s_mov_b64 s[%d:%d], exec" % (S_SAVE, S_SAVE + 1)]
    # (1) the accumulator was the point at infinity: acc <- (q, 1, 1)
    T_ += ["s_and_b64 exec, s[%d:%d], s[%d:%d]" % (S_SAVE, S_SAVE + 1, S_AI, S_AI + 1), "s_cbranch_execz .Lmb_r2%="]
    T_ += text_of_list(gen_first()) + ["v_mov_b32_e64 v%d, 0" % V_AINF]
    # (2) P == 0 on a finite accumulator: R == 0 -> the doubling of q, else the sum is the point at infinity
    T_ += [".Lmb_r2%=:", "s_andn2_b64 s[%d:%d], s[%d:%d], s[%d:%d]" % (S_ZP, S_ZP + 1, S_ZP, S_ZP + 1, S_AI, S_AI + 1),
           "s_and_b64 exec, s[%d:%d], s[%d:%d]" % (S_SAVE, S_SAVE + 1, S_ZP, S_ZP + 1), "s_cbranch_execz .Lmb_r9%="]
    T_ += text_of_list(gen_rtest())
    T_ += ["v_cmp_ne_u32_e64 vcc, 0, v%d" % V_ZP, "v_cndmask_b32_e64 v%d, v%d, 1, vcc" % (V_AINF, V_AINF),      # R != 0: the sum is infinity
           "v_cmp_eq_u32_e64 vcc, 0, v%d" % V_ZP, "s_and_b64 exec, exec, vcc", "s_cbranch_execz .Lmb_r9%="]
    T_ += text_of_list(gen_dbl())
    T_ += [".Lmb_r9%=:", "s_mov_b64 exec, s[%d:%d]" % (S_SAVE, S_SAVE + 1), "s_branch .Lmb_back%="]
    # ---- the bucket sum in Jacobian form, SoA store (three planes of four rows); an accumulator at infinity stores zeros
    T_ += [".Lmb_store%=:", "s_mov_b64 exec, s[%d:%d]" % (S_EXEC, S_EXEC + 1)]
    T_ += text_of_list(gen_final())
    T_ += ["v_cmp_ne_u32_e64 vcc, 0, v%d" % V_AINF, "s_and_saveexec_b64 s[%d:%d], vcc" % (S_T, S_T + 1)]
    T_ += ["v_mov_b32_e64 v%d, 0" % r for r in T[0].r + T[1].r + AZZ.r]
    T_ += ["s_mov_b64 exec, s[%d:%d]" % (S_T, S_T + 1)]
    T_ += ["s_mov_b64 s[%d:%d], s[%d:%d]" % (S_GP, S_GP + 1, S_OUT, S_OUT + 1)]
    for src in (T[0], T[1], AZZ):
        for q in range(4):
            r = src.r[4 * q]
            n = 4 if q < 3 else 2
            T_.append("global_store_dwordx%d v%d, v[%d:%d], s[%d:%d]" % (n, V_OFF, r, r + n - 1, S_GP, S_GP + 1))
            T_ += ["s_add_u32 s%d, s%d, s%d" % (S_GP, S_GP, S_OSTR), "s_addc_u32 s%d, s%d, 0" % (S_GP + 1, S_GP + 1)]
    T_ += ["s_waitcnt vmcnt(0)", "v_mov_b32_e64 %%0, v%d" % V_FLAG]
    return T_


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
    lines = [l if l.startswith(".L") else "\\t" + l for l in kernel_text_pipelined()]
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_msm_asm.py -- do not edit.\n"
           "// operands: %0 flag out (v), %1 point records (s pair), %2 %3 address of the lane's first index word (v, v), %4 entries (v), %5 bucket sums (s pair), %6 their row stride in bytes (s), %7 16 * output column (v)\n"
           "#define BLS_MSM_ASM_BODY \\\n" + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_MSM_ASM_CLOBBERS " + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
