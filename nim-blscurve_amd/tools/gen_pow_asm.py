#!/usr/bin/env python3
"""Generates fp_recip_sqrt_pow's device body (csrc/fp.hpp): t = a^((p - 3) / 4) for ONE Fp element per lane, the exponentiation behind every square
root of the path - two per SSWU map of hash-to-G2 (h2c.hpp sqrt_ratio_fp2: 85 % of k_hash_map's multiply-adds; blst_abi.nim:383), one per
decompressed coordinate of the batched deserialisation (deser.hpp) - as ONE hand-allocated gfx950 inline-asm statement.

The exponent is the same for every lane of every wave, so nothing about the chain is data: a 4-bit sliding window (8 odd powers a, a^3 .. a^15;
379 squarings + 76 window products + 1 + 7 for the table = 84 products - the 5-bit window of the compiled form takes 67 + 16 = 83 with twice the
table) whose schedule is unrolled into the code: `n squarings, then times entry j` with n and j immediates.  The table therefore lives in
REGISTERS (v60 .. v171, copied into the product's operand slot where the schedule says so) instead of the scratch memory the compiled form indexes it in
(k_hash_map: 1 824 bytes of scratch per lane, 25 x its algorithmic bytes in HBM traffic, 155 spilled registers around the out-of-line calls),
the squaring and the product are two subroutines over fixed registers (the squaring's loop inside its subroutine: one taken branch per
squaring), and the whole statement needs 172 VGPRs and no AGPRs: two waves per SIMD as before.

`--selftest`: asmlib's interpreter runs the generated table preparation, squaring and product blocks through the whole schedule against
Python's pow (tests/test_asm_loops.py); the loop control of a run of squarings is raw text, covered by the GPU parity tests (every hash-to-G2
stage output and every decompressed point against the oracle)."""
import argparse
import random
import sys

from asmlib import Asm, Builder, Machine, MASK, N0, NL, P, PL, R, RECIP, RINV, blk, check_limbs, get, limbs_of, mmul, put

E_PM3D4 = (P - 3) // 4
W = 4
# ---- register plan ----------------------------------------------------------------------------------------------------------------------
RR, X2 = blk(0), blk(14)                       # the running power; the squaring's doubled operand
M_REGS = [28 + i for i in range(NL)]
ACC, TMP, TMP2 = 42, 44, 45
SLOT = blk(46)                                 # second operand of the product: a^2 while the table is built, then the window's entry
TAB = [blk(60 + 14 * i) for i in range(8)]     # a^1, a^3, .. a^15 in v60 .. v171 (VGPRs, not AGPRs: in the unified register file of a 256-register
CLOBBER_V = 172                                # kernel the AGPRs start behind the kernel's own VGPR allocation, so an AGPR table pushed k_hash_map to 376
CLOBBER_A = 0                                  # registers = one wave per SIMD)
S_P, S_N0, S_MASK, S_RECIP = 36, 50, 51, 52
S_SQR, S_MUL, S_RET, S_CNT = 54, 56, 58, 60
CLOBBER_S = (36, 62)
IN_BOUNDS = (8, 2)                             # what a caller may pass: |a| < 8 p, limbs of two units (sums of two carried values)


def sliding(e, w):
    """pairs (squarings, odd window value or 0), as tools/gen_constants.py"""
    ops, i, pend = [], e.bit_length() - 1, 0
    while i >= 0:
        if not (e >> i) & 1:
            pend += 1
            i -= 1
            continue
        lo = max(i - w + 1, 0)
        while not (e >> lo) & 1:
            lo += 1
        ops.append((pend + i - lo + 1, (e >> lo) & ((1 << (i - lo + 1)) - 1)))
        pend, i = 0, lo - 1
    if pend:
        ops.append((pend, 0))
    r = 0
    for n, v in ops:
        r = (r << n) + v
    assert r == e
    return ops


SCHED = sliding(E_PM3D4, W)


def new_asm():
    return Asm(ACC, S_P, S_N0, S_MASK, S_RECIP)


def builder(a):
    return Builder(a, M_REGS, TMP, TMP2)


def sub_sqr():
    """RR <- RR^2 (one squaring; the run's loop is around it in the text)"""
    a = new_asm(); b = builder(a)
    b.sqr(RR.like(2, 0), X2, RR)
    return a.ins


def sub_mul():
    """RR <- RR * SLOT"""
    a = new_asm(); b = builder(a)
    b.dot([(RR.like(2, 0), SLOT.like(2, 0))], RR)
    return a.ins


def gen_prep():
    """RR = a (caller's bounds) -> the table a, a^3 .. a^15 in TAB, SLOT = a^2"""
    a = new_asm(); b = builder(a)
    x = RR.like(*IN_BOUNDS)
    first = b.carry(RR, x) if IN_BOUNDS[1] > 1 else x                    # limbs of one unit: the multiplier takes two
    first = first.like(IN_BOUNDS[0], 1)
    b.mov(TAB[0], first)
    b.mov(SLOT, first)
    b.sqr(first, X2, RR)                                                 # a^2
    a2 = RR.like(2, 0)
    # a^3 = a^2 * a (SLOT still holds a), then SLOT = a^2 and every further entry is the previous one times a^2
    b.dot([(a2, SLOT.like(IN_BOUNDS[0], 1))], X2)                         # X2 is free between squarings
    b.mov(SLOT, a2)
    b.mov(RR, X2.like(2, 0))
    b.mov(TAB[1], RR.like(2, 0))
    for i in range(2, 8):
        b.dot([(RR.like(2, 0), SLOT.like(2, 0))], RR)
        b.mov(TAB[i], RR.like(2, 0))
    return a.ins


def load_entry(j, dst):
    a = new_asm(); b = builder(a)
    b.mov(dst, TAB[j].like(*(IN_BOUNDS[0], 1) if j == 0 else (2, 0)))
    return a.ins


def program():
    """[(kind, arg)]: ("ins", list) straight-line blocks, ("sqr", n) a run of n squarings, ("mul", j) times entry j"""
    out = [("ins", gen_prep()), ("ins", load_entry(SCHED[0][1] >> 1, RR))]
    for n, v in SCHED[1:]:
        out.append(("sqr", n))
        if v:
            out.append(("mul", v >> 1))
    return out


def selftest(seed=7):
    rnd = random.Random(seed)
    subs = {"SQR": sub_sqr(), "MUL": sub_mul()}
    nsq = sum(n for n, _ in SCHED[1:])
    nmul = sum(1 for _, v in SCHED[1:] if v)
    assert nsq == E_PM3D4.bit_length() - SCHED[0][1].bit_length()
    for trial in range(3):
        mach = Machine(new_asm(), subs)
        x = rnd.randrange(P) if trial else 1
        # the widest input the contract allows: a limb-wise sum of residues, |a| < 8 p
        parts = [rnd.randrange(P) for _ in range(2)] if trial == 2 else [x]
        if trial == 2:
            x = sum(parts) % P
        limbs = [sum(limbs_of(p_)[i] for p_ in parts) for i in range(NL)]
        put(mach, RR, limbs)
        c0 = dict(mach.count)
        for kind, arg in program():
            if kind == "ins":
                mach.run(arg)
            elif kind == "sqr":
                for _ in range(arg):
                    mach.run(subs["SQR"])
            else:
                mach.run(load_entry(arg, SLOT))
                mach.run(subs["MUL"])
        # Montgomery images: x = X R, product = a b / R, so the chain returns X^e R
        xe = pow(x * RINV % P, E_PM3D4, P) * R % P
        assert get(mach, RR) % P == xe, "a^((p-3)/4)"
        check_limbs(mach, RR, 0)
        assert abs(get(mach, RR)) < 2 * P
        tot = {k: mach.count[k] - c0[k] for k in c0}
    print("gen_pow_asm selftest ok: %d squarings + %d window products + 8 for the table; %d VALU instructions, %d multiply-adds (%.1f %%); squaring %d, product %d instructions"
          % (nsq, nmul, tot["valu"], tot["mad"], 100.0 * tot["mad"] / tot["valu"], len(subs["SQR"]), len(subs["MUL"])))


def text_of_list(ins):
    a = new_asm(); a.ins = ins
    return a.text()


def kernel_text():
    """operands %0 .. %13 (in/out, v): the limbs of a in, of a^((p-3)/4) out"""
    T = []
    T += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK), "s_mov_b32 s%d, 0x%x" % (S_RECIP, RECIP)]
    T += ["v_mov_b32_e64 v%d, %%%d" % (RR.r[i], i) for i in range(NL)]
    for name, sreg in (("sqr", S_SQR), ("mul", S_MUL)):
        T += ["s_getpc_b64 s[%d:%d]" % (sreg, sreg + 1), ".Lpw_p%s%%=:" % name,
              "s_add_u32 s%d, s%d, (.Lpw_%s%%=-.Lpw_p%s%%=)&4294967295" % (sreg, sreg, name, name),
              "s_addc_u32 s%d, s%d, (.Lpw_%s%%=-.Lpw_p%s%%=)>>32" % (sreg + 1, sreg + 1, name, name)]
    T += ["s_branch .Lpw_main%="]
    # a run of s_cnt squarings: the loop is inside the subroutine (one taken branch per squaring, the return at the end of the run)
    T += [".Lpw_sqr%=:"] + text_of_list(sub_sqr()) + ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 .Lpw_sqr%=",
                                                      "s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1)]
    T += [".Lpw_mul%=:"] + text_of_list(sub_mul()) + ["s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1)]
    T += [".Lpw_main%=:"]
    for kind, arg in program():
        if kind == "ins":
            T += text_of_list(arg)
        elif kind == "sqr":
            T += ["s_mov_b32 s%d, %d" % (S_CNT, arg), "s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_RET, S_RET + 1, S_SQR, S_SQR + 1)]
        else:
            T += text_of_list(load_entry(arg, SLOT)) + ["s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_RET, S_RET + 1, S_MUL, S_MUL + 1)]
    T += ["v_mov_b32_e64 %%%d, v%d" % (i, RR.r[i]) for i in range(NL)]
    return T


def clobbers():
    c = ["v%d" % i for i in range(CLOBBER_V)] + ["a%d" % i for i in range(CLOBBER_A)] + ["s%d" % i for i in range(*CLOBBER_S)] + ["vcc", "scc"]
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
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_pow_asm.py -- do not edit.\n"
           "// operands %0 .. %13 (in/out, v): the limbs of a in, of a^((p-3)/4) out\n"
           "#define BLS_POW_ASM_BODY \\\n" + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_POW_ASM_CLOBBERS " + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
