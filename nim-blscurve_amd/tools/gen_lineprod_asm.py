#!/usr/bin/env python3
"""Generates the hand-allocated gfx950 inner loop of k_lineprod (csrc/kernels.hip) as ONE inline-asm statement.

What it computes: per lane, f <- f * line_j for the lines j = 0 .. rounds-1 of one Miller step (line 0 initialises f), with the
SCHOOLBOOK product of tower.hpp's fp12_mul_by_line_lazy: every Fp2 coefficient of the result is a sum of three Fp2 products =
two Montgomery dot products of six Fp terms with one reduction each (fp.hpp fp_dotn_core<6>), 12 x (6 x 196 + 196) = 16 464
multiply-adds per line.

Why assembly: at one wave per SIMD (an Fp12 accumulator is 168 registers) nothing hides a wave's own stalls, and hipcc's code for
this loop stalls a lot - register spills reloaded right in front of their use (each a full memory latency), the next line's loads
waited for where they are issued, 56 argument moves and two instruction-fetch bubbles (~70 cycles each, tools/ubench_dep.hip) per
out-of-line multiplier call: 11-20 % of the wave's cycles are waits (profiles/r04_ab).  Here every value has a fixed home:
  v0..v83     the line's three Fp2 coefficients (14 limbs per Fp)
  v84..v125   the negated imaginary parts of the line (the real part of an Fp2 product subtracts im * im)
  v126..v209  three Fp2 operand slots, copied from the accumulator for the coefficient being computed
  v210..v223  the 14 Montgomery quotient digits m_k; v[224:225] the column accumulator; v226..v239 the real part of the coefficient
              being computed (its imaginary part lands in the first operand slot, whose limbs are dead by then); v240.. addresses
  a0..a251    nine blocks of 28: the six Fp2 coefficients of f and three blocks that take new coefficients while old ones are read
  SGPRs       the limbs of p, -1/p mod 2^28, the limb mask, the loop state
no scratch, no spills.  The two dot products of one coefficient (2 880 instructions, 23 KB) are ONE subroutine called six times per
line with its operands in place: the hot code is 33 KB.  (Fully unrolled, a line was 146 KB of straight code, and a loop body that
does not fit the 64 KB instruction cache costs ~4.6 cycles per instruction instead of ~4.1-4.3: tools/ubench_icache.hip.)

Output: csrc-includable text, one C string literal per instruction (build.sh writes it to build/lineprod_asm.inc).
`--selftest` executes the generated instruction list for one lane in a small interpreter and checks f * line against big-integer
arithmetic (no GPU needed); tests/test_lineprod_asm.py runs it.
"""
import argparse
import random
import sys

X_ABS = 0xd201000000010000
X = -X_ABS
P = (X - 1) ** 2 * (X ** 4 - X ** 2 + 1) // 3 + X
LB, NL = 28, 14
MASK = (1 << LB) - 1
R = 1 << (LB * NL)
N0 = (-pow(P, -1, 1 << LB)) % (1 << LB)
PL = [(P >> (LB * i)) & MASK for i in range(NL)]
ONE = [((R % P) >> (LB * i)) & MASK for i in range(NL)]

# ---- register plan -------------------------------------------------------------------------------------------------------------
L_BASE, NL_BASE, X_BASE, M_BASE, ACC, R_BASE, TMP = 0, 84, 126, 210, 224, 226, 240
V_OFF, V_OFF2, V_IDX, V_T = 242, 243, 244, 245          # line offset (bytes), output offset, pair index of this lane, scratch
V_AD, V_LDS = 246, 248                                  # 64-bit address of the LDS-DMA loads (pair), this lane's byte address in the DMA buffer
NBLK = 9
S_P, S_N0, S_MASK = 40, 54, 55                         # s40..s53 limbs of p
S_CNT, S_PC, S_T = 56, 58, 60                          # loop counter, saved loop-top pc (pair), scalar temporaries (s60..s63)
S_BASE, S_STR, S_NEXT = 64, 66, 68                     # step base (pair), plane stride in bytes as a 64-bit pair, exec mask of the NEXT line (pair)
S_LAST, S_SUB, S_RET = 70, 72, 74                      # lines left; address of the coefficient subroutine (pair); its return address (pair)
CLOBBER_V = 250                                        # v0..v249 are ours
# coefficient order inside f: a0 a1 a2 b0 b1 b2 (tower slots c0.a0 c0.a1 c0.a2 c1.a0 c1.a1 c1.a2)
A0, A1, A2, B0, B1, B2 = range(6)


def lreg(c, part, i):
    return L_BASE + 28 * c + 14 * part + i


def nlreg(c, i):
    return NL_BASE + 14 * c + i


def xreg(slot, part, i):
    return X_BASE + 28 * slot + 14 * part + i


class Gen:
    def __init__(self):
        self.ins = []          # (mnemonic, operands...) tuples; text is produced from them, and the interpreter executes them

    def e(self, *t):
        self.ins.append(t)

    # -- one Montgomery dot product of the pairs [(xv, yv)] (lists of 14 VGPR numbers each) -> result limbs to the VGPRs dst[0..13]
    #    (dst[k] may be a register that is dead from column 14 + k on: an m_k, or limb k of an operand whose pairs come first)
    def dot(self, pairs, dst):
        first = True
        for kk in range(2 * NL - 1):
            lo, hi = max(0, kk - NL + 1), min(kk, NL - 1)
            for xv, yv in pairs:
                for i in range(lo, hi + 1):
                    self.e("mad", xv[i], ("v", yv[kk - i]), first)
                    first = False
            if kk < NL:
                for i in range(kk):
                    self.e("mad", M_BASE + i, ("s", S_P + kk - i), False)
                self.e("mul_lo", M_BASE + kk, ACC, S_N0)
                self.e("and", M_BASE + kk, S_MASK, M_BASE + kk)
                self.e("mad", M_BASE + kk, ("s", S_P), False)
            else:
                for i in range(kk - NL + 1, NL):
                    self.e("mad", M_BASE + i, ("s", S_P + kk - i), False)
                self.e("and", dst[kk - NL], S_MASK, ACC)
            self.e("ashr28")
        self.e("mov", dst[NL - 1], ACC)

    def coefficient_body(self):
        """the subroutine: (X, Y, Z) in the operand slots times (l0, l1, l2): real part -> v[R_BASE..], imaginary part -> slot 0's real limbs"""
        re_pairs, im_pairs = [], []
        for s_ in range(3):
            xr = [xreg(s_, 0, i) for i in range(NL)]
            xi_ = [xreg(s_, 1, i) for i in range(NL)]
            lr = [lreg(s_, 0, i) for i in range(NL)]
            li = [lreg(s_, 1, i) for i in range(NL)]
            nli = [nlreg(s_, i) for i in range(NL)]
            re_pairs += [(xr, lr), (xi_, nli)]
            im_pairs += [(xr, li), (xi_, lr)]
        self.dot(re_pairs, [R_BASE + i for i in range(NL)])
        self.dot(im_pairs, [xreg(0, 0, i) for i in range(NL)])

    def load_slot(self, slot, blk, xi):
        """operand slot <- coefficient in AGPR block blk (xi: times 1 + u, limb-wise: (re - im, re + im))"""
        for i in range(NL):
            if not xi:
                self.e("aread", xreg(slot, 0, i), 28 * blk + i)
                self.e("aread", xreg(slot, 1, i), 28 * blk + 14 + i)
            else:
                self.e("aread", TMP, 28 * blk + i)
                self.e("aread", TMP + 1, 28 * blk + 14 + i)
                self.e("sub", xreg(slot, 0, i), TMP, TMP + 1)
                self.e("add", xreg(slot, 1, i), TMP, TMP + 1)

    def output(self, target_blk, ops):
        """new coefficient = sum over slot s of ops[s] * l_s; ops[s] = (block, xi)"""
        for s_, (blk, xi) in enumerate(ops):
            self.load_slot(s_, blk, xi)
        self.e("call")
        for i in range(NL):
            self.e("awrite", 28 * target_blk + i, R_BASE + i)
        for i in range(NL):
            self.e("awrite", 28 * target_blk + 14 + i, xreg(0, 0, i))

    def neg_line_im(self):
        for c in range(3):
            for i in range(NL):
                self.e("neg", nlreg(c, i), lreg(c, 1, i))

    def product(self):
        """f <- f * line; f's coefficients a0 a1 a2 b0 b1 b2 live in blocks 0..5 before and after; blocks 6..8 are free"""
        self.neg_line_im()
        self.output(6, [(A0, False), (A2, True), (B1, True)])      # c0.a0 = a0 l0 + xi a2 l1 + xi b1 l2
        self.output(7, [(A1, False), (A0, False), (B2, True)])     # c0.a1 = a1 l0 +    a0 l1 + xi b2 l2
        self.output(8, [(B1, False), (B0, False), (A0, False)])    # c1.a1 = b1 l0 +    b0 l1 +    a0 l2      (a0 is dead now)
        self.output(0, [(A2, False), (A1, False), (B0, False)])    # c0.a2 = a2 l0 +    a1 l1 +    b0 l2      -> block 0
        self.output(3, [(B0, False), (B2, True), (A2, True)])      # c1.a0 = b0 l0 + xi b2 l1 + xi a2 l2      in place (b0, a2 dead)
        self.output(5, [(B2, False), (B1, False), (A1, False)])    # c1.a2 = b2 l0 +    b1 l1 +    a1 l2      in place
        for i in range(28):                                        # new a2: block 0 -> 2; new a0, a1, b1: blocks 6, 7, 8 -> 0, 1, 4
            self.e("amov", 28 * 2 + i, 28 * 0 + i)
        for src, dstb in ((6, 0), (7, 1), (8, 4)):
            for i in range(28):
                self.e("amov", 28 * dstb + i, 28 * src + i)

    def from_line(self):
        """f <- the line itself: a0 = l0, a1 = l1, b1 = l2, the rest 0 (lanes outside exec keep f = 1)"""
        for c, blk in ((0, A0), (1, A1), (2, B1)):
            for part in range(2):
                for i in range(NL):
                    self.e("awrite", 28 * blk + 14 * part + i, lreg(c, part, i))
        self.e("movi", TMP, 0)
        for blk in (A2, B0, B2):
            for i in range(28):
                self.e("awrite", 28 * blk + i, TMP)

    def init_one(self):
        for i in range(NL):
            self.e("movi", TMP, ONE[i])
            self.e("awrite", i, TMP)
        self.e("movi", TMP, 0)
        for a in range(NL, 28 * 6):
            self.e("awrite", a, TMP)


# ---- text ----------------------------------------------------------------------------------------------------------------------
def text_of(t):
    op = t[0]
    if op == "mad":
        _, x, (kind, y), first = t
        ysrc = ("v%d" % y) if kind == "v" else ("s%d" % y)
        add = "0" if first else "v[%d:%d]" % (ACC, ACC + 1)
        return "v_mad_i64_i32 v[%d:%d], vcc, v%d, %s, %s" % (ACC, ACC + 1, x, ysrc, add)
    if op == "mul_lo":
        return "v_mul_lo_u32 v%d, v%d, s%d" % (t[1], t[2], t[3])
    if op == "and":
        return "v_and_b32_e64 v%d, s%d, v%d" % (t[1], t[2], t[3])
    if op == "ashr28":
        return "v_ashrrev_i64 v[%d:%d], 28, v[%d:%d]" % (ACC, ACC + 1, ACC, ACC + 1)
    if op == "awrite":
        return "v_accvgpr_write_b32 a%d, v%d" % (t[1], t[2])
    if op == "aread":
        return "v_accvgpr_read_b32 v%d, a%d" % (t[1], t[2])
    if op == "amov":
        return "v_accvgpr_mov_b32 a%d, a%d" % (t[1], t[2])
    if op == "sub":
        return "v_sub_u32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
    if op == "add":
        return "v_add_u32_e64 v%d, v%d, v%d" % (t[1], t[2], t[3])
    if op == "neg":
        return "v_sub_u32_e64 v%d, 0, v%d" % (t[1], t[2])
    if op == "movi":
        return "v_mov_b32_e32 v%d, 0x%x" % (t[1], t[2])
    if op == "mov":
        return "v_mov_b32_e64 v%d, v%d" % (t[1], t[2])                 # 8-byte encoding: the stream stays 8-byte aligned without the post-pass too
    if op == "call":
        return "s_swappc_b64 s[%d:%d], s[%d:%d]" % (S_RET, S_RET + 1, S_SUB, S_SUB + 1)
    if op == "raw":
        return t[1]
    raise ValueError(op)


def dma_issue_text():
    """The NEXT line (byte offset v[V_OFF] + 1024 from the step base) straight into the LDS buffer: 24 x global_load_lds_dwordx4, one
    1 KiB row per (plane, limb group) in memory order, no VGPR destination - the loads stay in flight while the current line is
    multiplied.  Lanes whose next pair does not exist take no part (exec = s[S_NEXT])."""
    out = ["s_mov_b64 exec, s[%d:%d]" % (S_NEXT, S_NEXT + 1),
           "v_add_u32_e32 v%d, 0x400, v%d" % (V_T, V_OFF),
           "v_mov_b32_e32 v%d, 0" % (V_AD + 1),
           "v_mov_b32_e32 v%d, v%d" % (V_AD, V_T),
           "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (V_AD, V_AD + 1, V_AD, V_AD + 1, S_BASE, S_BASE + 1),
           "s_mov_b32 m0, %6"]
    for k in range(24):
        out += ["s_nop 0", "global_load_lds_dwordx4 v[%d:%d], off" % (V_AD, V_AD + 1)]
        if k < 23:
            out += ["v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (V_AD, V_AD + 1, V_AD, V_AD + 1, S_STR, S_STR + 1), "s_add_u32 m0, m0, 0x400"]
    return out


def lds_fetch_text():
    """the line that the last dma_issue_text brought in: LDS (its memory image, 16 words per Fp) -> v0..v83 (14 limbs per Fp)"""
    out = ["s_waitcnt vmcnt(0)"]
    for c in range(3):
        for part in range(2):
            k = 4 * (2 * c + part)
            for q in range(3):
                r = lreg(c, part, 4 * q)
                out.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (r, r + 3, V_LDS, 1024 * (k + q)))
            r = lreg(c, part, 12)
            out.append("ds_read_b64 v[%d:%d], v%d offset:%d" % (r, r + 1, V_LDS, 1024 * (k + 3)))
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def next_mask_text():
    """s[S_NEXT] <- lanes whose pair of the next line exists (index + 64 < npairs), all zero behind the last round"""
    return ["v_add_u32_e64 v%d, 64, v%d" % (V_T, V_IDX), "v_cmp_gt_u32_e64 s[%d:%d], %%2, v%d" % (S_NEXT, S_NEXT + 1, V_T),
            "s_cmp_lt_i32 s%d, 1" % S_LAST, "s_cselect_b32 s%d, 0, s%d" % (S_NEXT, S_NEXT), "s_cselect_b32 s%d, 0, s%d" % (S_NEXT + 1, S_NEXT + 1)]


def kernel_text():
    """the whole statement: operands %0 step base (s pair), %1 stride16 (s), %2 npairs (s), %3 first (s), %4 rounds (s), %5 out (s pair),
    %6 byte address of a 24 KiB LDS buffer (s)"""
    T = []
    T += ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK)]
    T += ["s_mov_b64 s[%d:%d], %%0" % (S_BASE, S_BASE + 1), "s_mov_b32 s%d, %%1" % S_STR, "s_mov_b32 s%d, 0" % (S_STR + 1)]
    T += ["v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_T, "v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_T, V_T)]                   # lane id
    T += ["v_mul_u32_u24_e32 v%d, 0x300, v%d" % (V_OFF2, V_T)]                                                     # 768 bytes of output per lane
    T += ["v_lshlrev_b32_e64 v%d, 4, v%d" % (V_LDS, V_T), "v_add_u32_e64 v%d, %%6, v%d" % (V_LDS, V_LDS)]
    T += ["v_add_u32_e64 v%d, %%3, v%d" % (V_IDX, V_T), "v_lshlrev_b32_e64 v%d, 4, v%d" % (V_OFF, V_IDX)]
    g = Gen(); g.init_one(); T += [text_of(t) for t in g.ins]
    # the coefficient subroutine sits in front of the loop that calls it; s[S_SUB] = its address
    T += ["s_getpc_b64 s[%d:%d]" % (S_SUB, S_SUB + 1), ".Llp_sp%=:",
          "s_add_u32 s%d, s%d, (.Llp_sub%%=-.Llp_sp%%=)&4294967295" % (S_SUB, S_SUB),
          "s_addc_u32 s%d, s%d, (.Llp_sub%%=-.Llp_sp%%=)>>32" % (S_SUB + 1, S_SUB + 1),
          "s_getpc_b64 s[%d:%d]" % (S_T + 2, S_T + 3), ".Llp_mp%=:",
          "s_add_u32 s%d, s%d, (.Llp_main%%=-.Llp_mp%%=)&4294967295" % (S_T + 2, S_T + 2),
          "s_addc_u32 s%d, s%d, (.Llp_main%%=-.Llp_mp%%=)>>32" % (S_T + 3, S_T + 3),
          "s_setpc_b64 s[%d:%d]" % (S_T + 2, S_T + 3), ".Llp_sub%=:"]
    g = Gen(); g.coefficient_body(); T += [text_of(t) for t in g.ins]
    T += ["s_setpc_b64 s[%d:%d]" % (S_RET, S_RET + 1), ".Llp_main%=:"]
    # line 0: bring it in (the "next line" of a position one line before the first), then f <- line 0 under exec = (pair index < npairs)
    T += ["v_subrev_u32_e32 v%d, 64, v%d" % (V_IDX, V_IDX), "v_subrev_u32_e32 v%d, 0x400, v%d" % (V_OFF, V_OFF), "s_mov_b32 s%d, %%4" % S_LAST]
    T += next_mask_text() + dma_issue_text()
    T += ["s_mov_b64 exec, -1", "v_add_u32_e64 v%d, 64, v%d" % (V_IDX, V_IDX), "v_add_u32_e32 v%d, 0x400, v%d" % (V_OFF, V_OFF)]
    T += ["s_sub_u32 s%d, %%4, 1" % S_CNT, "s_mov_b32 s%d, s%d" % (S_LAST, S_CNT)]              # lines left behind line 0
    T += ["v_cmp_gt_u32_e64 vcc, %%2, v%d" % V_IDX, "s_and_saveexec_b64 s[%d:%d], vcc" % (S_T, S_T + 1)]
    T += lds_fetch_text()
    T += ["s_mov_b64 s[%d:%d], exec" % (S_T, S_T + 1)] + next_mask_text() + dma_issue_text() + ["s_mov_b64 exec, s[%d:%d]" % (S_T, S_T + 1)]
    g = Gen(); g.from_line(); T += [text_of(t) for t in g.ins]
    T += ["s_mov_b64 exec, -1"]
    # lines 1 .. rounds - 1
    T += ["s_cmp_lt_i32 s%d, 1" % S_CNT, "s_cbranch_scc1 .Llp_end%="]
    T += ["s_getpc_b64 s[%d:%d]" % (S_PC, S_PC + 1)]                                                 # = address of the next instruction
    T += ["v_add_u32_e64 v%d, 64, v%d" % (V_IDX, V_IDX), "v_add_u32_e32 v%d, 0x400, v%d" % (V_OFF, V_OFF)]
    T += ["s_sub_u32 s%d, s%d, 1" % (S_LAST, S_CNT)]                                                                  # lines left behind this one
    T += ["v_cmp_gt_u32_e64 vcc, %%2, v%d" % V_IDX, "s_and_saveexec_b64 s[%d:%d], vcc" % (S_T, S_T + 1)]
    T += lds_fetch_text()
    T += ["s_mov_b64 s[%d:%d], exec" % (S_T, S_T + 1)] + next_mask_text() + dma_issue_text() + ["s_mov_b64 exec, s[%d:%d]" % (S_T, S_T + 1)]
    g = Gen(); g.product(); T += [text_of(t) for t in g.ins]
    T += ["s_mov_b64 exec, -1", "s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_lt_i32 s%d, 1" % S_CNT, "s_cbranch_scc1 .Llp_end%=",
          "s_setpc_b64 s[%d:%d]" % (S_PC, S_PC + 1), ".Llp_end%=:"]
    # store f: 12 Fp x 16 words (st_fp12_int layout), 768 bytes per lane at %5 + lane * 768
    for c in range(6):
        for part in range(2):
            for q in range(4):
                a = 28 * c + 14 * part + 4 * q
                T.append("global_store_dwordx4 v%d, a[%d:%d], %%5 offset:%d" % (V_OFF2, a, a + 3, (2 * c + part) * 64 + 16 * q))
    T += ["s_waitcnt vmcnt(0)"]
    return T


def clobbers():
    # m0 (LDS-DMA base; LLVM refuses it on a clobber list: a reserved register) and scc (s_cmp / s_sub / s_cselect) are written too; exec is changed (saveexec / s_mov exec) and left at -1: the statement
    # is the LAST thing of its kernel branch (k_lineprod returns right behind it) - nothing may be placed after it that assumes m0 or exec
    c = ["v%d" % i for i in range(CLOBBER_V)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in range(S_P, S_RET + 2)] + ["vcc", "scc", "memory"]
    return ", ".join('"%s"' % x for x in c)


# ---- one-lane interpreter (selftest) -------------------------------------------------------------------------------------------
def s32(x):
    x &= 0xffffffff
    return x - (1 << 32) if x >> 31 else x


def run(ins, v, a, s):
    acc = 0
    sub = Gen(); sub.coefficient_body()
    flat = []
    for t in ins:
        flat += sub.ins if t[0] == "call" else [t]
    for t in flat:
        op = t[0]
        if op == "mad":
            _, x, (kind, y), first = t
            yv = v[y] if kind == "v" else s[y]
            acc = (0 if first else acc) + s32(v[x]) * s32(yv)
            assert -(1 << 63) <= acc < (1 << 63), "column accumulator overflow"
            v[ACC], v[ACC + 1] = acc & 0xffffffff, (acc >> 32) & 0xffffffff
        elif op == "mul_lo":
            v[t[1]] = (v[t[2]] * s[t[3]]) & 0xffffffff
        elif op == "and":
            v[t[1]] = s[t[2]] & v[t[3]]
        elif op == "ashr28":
            acc >>= 28
            v[ACC], v[ACC + 1] = acc & 0xffffffff, (acc >> 32) & 0xffffffff
        elif op == "awrite":
            a[t[1]] = v[t[2]]
        elif op == "aread":
            v[t[1]] = a[t[2]]
        elif op == "amov":
            a[t[1]] = a[t[2]]
        elif op == "sub":
            v[t[1]] = (v[t[2]] - v[t[3]]) & 0xffffffff
        elif op == "add":
            v[t[1]] = (v[t[2]] + v[t[3]]) & 0xffffffff
        elif op == "neg":
            v[t[1]] = (-v[t[2]]) & 0xffffffff
        elif op == "movi":
            v[t[1]] = t[2]
        elif op == "mov":
            v[t[1]] = v[t[2]]
        else:
            raise ValueError(op)


def limbs_of(x):
    return [(x >> (LB * i)) & MASK for i in range(NL)]


def value_of(ls):
    return sum(s32(l) << (LB * i) for i, l in enumerate(ls))


def selftest(rounds=3):
    rnd = random.Random(7)
    s = {S_P + i: PL[i] for i in range(NL)}
    s[S_N0], s[S_MASK] = N0, MASK
    v, a = [0] * 256, [0] * 256
    # f = random Fp12 (Montgomery images as canonical limbs), line = random, with one negated-limb coefficient as the addition steps store
    f = [[rnd.randrange(P), rnd.randrange(P)] for _ in range(6)]
    for c in range(6):
        for part in range(2):
            for i, l in enumerate(limbs_of(f[c][part])):
                a[28 * c + 14 * part + i] = l
    xi = lambda z: ((z[0] - z[1]) % P, (z[0] + z[1]) % P)
    mul = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    add3 = lambda x, y, z: ((x[0] + y[0] + z[0]) % P, (x[1] + y[1] + z[1]) % P)
    rinv = pow(R, -1, P)
    for it in range(rounds):
        line = [[rnd.randrange(P), rnd.randrange(P)] for _ in range(3)]
        for c in range(3):
            for part in range(2):
                ls = limbs_of(line[c][part])
                if c == 1 and it % 2 == 1:
                    ls = [(-x) & 0xffffffff for x in limbs_of((P - line[c][part]) % P)]          # the value stored as limb-wise negation
                for i, l in enumerate(ls):
                    v[lreg(c, part, i)] = l
        g = Gen(); g.product()
        run(g.ins, v, a, s)
        a0, a1, a2, b0, b1, b2 = [tuple(x) for x in f]
        l0, l1, l2 = [tuple(x) for x in line]
        exp = [add3(mul(a0, l0), mul(xi(a2), l1), mul(xi(b1), l2)), add3(mul(a1, l0), mul(a0, l1), mul(xi(b2), l2)),
               add3(mul(a2, l0), mul(a1, l1), mul(b0, l2)), add3(mul(b0, l0), mul(xi(b2), l1), mul(xi(a2), l2)),
               add3(mul(b1, l0), mul(b0, l1), mul(a0, l2)), add3(mul(b2, l0), mul(b1, l1), mul(a1, l2))]
        for c in range(6):
            for part in range(2):
                ls = [a[28 * c + 14 * part + i] for i in range(NL)]
                got = value_of(ls)
                assert all(0 <= s32(l) < (1 << LB) for l in ls[:13]), "limbs not canonical"
                assert abs(got) < 2 * P, "value out of (-2p, 2p)"
                assert got % P == exp[c][part] * rinv % P, ("coefficient", c, part, "round", it)
                f[c][part] = got % P
    # from_line / init_one
    g = Gen(); g.init_one(); run(g.ins, v, a, s)
    assert value_of([a[i] for i in range(NL)]) == R % P and all(x == 0 for x in a[NL:168])
    g = Gen(); g.from_line(); run(g.ins, v, a, s)
    assert [a[28 * A1 + i] for i in range(NL)] == [v[lreg(1, 0, i)] for i in range(NL)] and all(a[28 * B0 + i] == 0 for i in range(28))
    g = Gen(); g.product()
    sub = Gen(); sub.coefficient_body()
    n = len(g.ins) + 6 * (len(sub.ins) + 1)
    mads = 6 * sum(1 for t in sub.ins if t[0] == "mad")
    print("gen_lineprod_asm selftest ok: %d instructions per line (%d in the caller, %d in the subroutine), %d multiply-adds (%.1f %%)"
          % (n, len(g.ins), len(sub.ins), mads, 100.0 * mads / n))


def ubench_text(bodies=4):
    """tools/ubench_fp2chain.hip: `bodies` lazily reduced dot products a0 b0 + a1 b1 (fp.hpp fp_dot2_core: 588 multiply-adds + 68
    bookkeeping instructions each, half an Fp2 product) back to back with NO caller code at all - the instruction mix the field
    multipliers cannot do better than; every body's result feeds the next one's operand so that nothing can be dropped"""
    global L_BASE, X_BASE, M_BASE, ACC
    keep = (L_BASE, X_BASE, M_BASE, ACC)
    T = ["s_mov_b32 s%d, 0x%x" % (S_P + i, PL[i]) for i in range(NL)]
    T += ["s_mov_b32 s%d, 0x%x" % (S_N0, N0), "s_mov_b32 s%d, 0x%x" % (S_MASK, MASK)]
    g = Gen()
    xa, ya, xb, yb = ([b + i for i in range(NL)] for b in (0, 14, 28, 42))
    M_BASE, ACC = 56, 70
    for k in range(bodies):
        g.dot([(xa, ya), (xb, yb)], xa if k % 2 else xb)
    T += [text_of(t) for t in g.ins]
    L_BASE, X_BASE, M_BASE, ACC = keep
    return T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--ubench-dot2", type=int, default=0, help="emit N back-to-back dot2 bodies (tools/ubench_fp2chain.hip) instead of the kernel loop")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    if a.selftest:
        selftest()
        return
    if a.ubench_dot2:
        lines = ["\\t" + l for l in ubench_text(a.ubench_dot2)]
        txt = "// GENERATED by nim-blscurve_amd/tools/gen_lineprod_asm.py --ubench-dot2 %d\n#define UBENCH_BODY \\\n" % a.ubench_dot2
        txt += "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
        open(a.out, "w").write(txt) if a.out else sys.stdout.write(txt)
        return
    lines = [l if l.startswith(".L") else "\\t" + l for l in kernel_text()]      # instructions indented: tools/align_isa.py recognises them that way
    txt = ("// GENERATED by nim-blscurve_amd/tools/gen_lineprod_asm.py -- do not edit.\n"
           "// operands: %%0 step base (s pair), %%1 stride16 (s), %%2 npairs (s), %%3 first (s), %%4 rounds (s), %%5 out (s pair), %%6 LDS buffer (s)\n"
           "#define BLS_LINEPROD_ASM_BODY \\\n" + "\n".join('    "%s\\n" \\' % l for l in lines) + "\n\n"
           "#define BLS_LINEPROD_ASM_CLOBBERS " + clobbers() + "\n")
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
