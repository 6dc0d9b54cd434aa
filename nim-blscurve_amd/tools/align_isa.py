#!/usr/bin/env python3
"""Post-pass over the gfx950 assembly hipcc emits for csrc/kernels.hip: keep every 8-byte instruction 8-byte aligned.

Why: on gfx950 a stream of 8-byte VALU instructions (v_mad_u64_u32 / v_mad_i64_i32 are VOP3, 8 bytes) issues one
instruction per 4.3 cycles when the stream is 8-byte aligned and one per 5.3 cycles when it sits at offset 4
(tools/ubench_align.hip, profiles/r01_ubench_align.txt): every instruction then straddles a fetch boundary.  The
compiler aligns functions to 4 bytes and freely mixes 4-byte encodings (VOP1/VOP2/VOPC "_e32", scalar ALU) into
the stream, so about half of the multiply-adds of the field multipliers ran misaligned.

What it does, on the text of the .s file:
  0. hipcc puts one wait state (`s_nop 0`) behind every inline-asm statement whose result the next instruction reads: it cannot
     see what the asm is and assumes an instruction with destination selection (the gfx940 dst_sel forwarding hazard).  The only
     asm statements of this library are single v_mad_i64_i32 (fp.hpp bls_mac: the running column sum as the multiply-add's own
     addend); hipcc's own v_mad_i64_i32 chains carry no wait states, so those nops are removed (each costs an issue slot);
  1. every VOP1/VOP2/VOPC instruction in its 4-byte "_e32" encoding is re-encoded as VOP3 "_e64" (8 bytes, same
     operation); the few forms the assembler rejects (literal operands stay VOP2 + literal = 8 bytes anyway) are
     put back;
  2. functions start 8-byte aligned (.p2align 2 -> .p2align 3);
  3. the remaining 4-byte instructions are scalar (s_mov_b32, s_waitcnt, s_add_u32, branches, ...): wherever an
     8-byte instruction would land at offset 4, one `s_nop 0` (4 bytes) is inserted in front of it (before its
     label, if it has one, so that a jump to the label lands on the aligned instruction).
The result is assembled and the alignment of every 8-byte instruction is verified from the disassembly.

usage: align_isa.py in.s out.s [--clang CLANG] [--objdump OBJDUMP]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

INSTR = re.compile(r"^\s+([a-z][a-z0-9_]*)(\s|$)")
LABEL = re.compile(r"^[.\w$]+:")
INLINE_INT = range(-16, 65)
INLINE_FLOATS = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494", "0.15915494309189532"}
EIGHT_BYTE_PREFIX = ("ds_", "global_", "scratch_", "flat_", "buffer_", "tbuffer_", "image_", "s_load_", "s_store_", "s_buffer_",
                     "s_scratch_", "s_memtime", "s_memrealtime", "s_dcache_", "s_atc_", "s_atomic_", "v_accvgpr_", "v_mfma_", "v_smfmac_")


def is_instr(line):
    m = INSTR.match(line)
    if not m:
        return None
    mn = m.group(1)
    if mn.startswith(("v_", "s_", "ds_", "global_", "scratch_", "flat_", "buffer_", "tbuffer_", "image_")):
        return mn
    return None


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def has_literal(line):
    """True when an operand needs a 32-bit literal dword after the instruction."""
    for o in operands(line):
        t = o.split()[0] if o else o
        if not t:
            continue
        if re.match(r"^(v|s|a|ttmp)\d+$", t) or re.match(r"^(v|s|a|ttmp)\[", t) or t in ("vcc", "vcc_lo", "vcc_hi", "exec", "exec_lo", "exec_hi", "m0", "scc", "off", "null", "src_scc", "src_vccz", "src_execz", "src_shared_base", "src_shared_limit", "src_private_base", "src_private_limit", "src_pops_exiting_wave_id"):
            continue
        if t.startswith("("):
            return True                        # relocatable expression
        if re.match(r"^-?(0x[0-9a-fA-F]+|\d+)$", t):
            v = int(t, 0)
            if v in INLINE_INT:
                continue
            return True
        if t in INLINE_FLOATS:
            continue
        if re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", t):
            return True
        if re.match(r"^[A-Za-z_.$][\w.$@+\-]*$", t) and not re.match(r"^(offset|offset0|offset1|gds|glc|slc|nt|sc0|sc1|lds|dlc|tfe|idxen|offen|addr64|vmcnt|lgkmcnt|expcnt|row_\w+|quad_perm|bank_mask|row_mask|bound_ctrl|op_sel\w*|neg_\w+|clamp|mul|div|dst_sel|src0_sel|src1_sel|dst_unused)", t):
            return True                        # symbol / expression (e.g. sym@rel32@lo+4)
    return False


FOUR_BYTE = {"v_readfirstlane_b32", "v_nop", "v_swap_b32", "v_clrexcp", "v_accvgpr_mov_b32"}      # VOP1 printed without an encoding suffix


def size_of(line, mn):
    if mn in FOUR_BYTE:
        return 4
    if mn.startswith(EIGHT_BYTE_PREFIX):
        return 8
    if mn.startswith("v_"):
        if mn.endswith("_e32"):
            return 8 if has_literal(line) else 4
        if mn.endswith(("_e64", "_dpp", "_sdwa")):
            return 8
        return 8                               # VOP3 / VOP3P without suffix
    if mn.startswith("s_"):
        sopp = ("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_sethalt", "s_trap",
                "s_icache_inv", "s_incperflevel", "s_decperflevel", "s_ttracedata", "s_code_end", "s_sendmsg", "s_setkill", "s_wakeup",
                "s_set_gpr_idx_off", "s_set_gpr_idx_mode", "s_endpgm_saved", "s_endpgm_ordered_ps_done")
        if mn.startswith(sopp):
            return 4
        if mn.startswith(("s_movk_", "s_cmovk_", "s_cmpk_", "s_addk_", "s_mulk_", "s_getreg_", "s_call_")):
            return 4
        if mn.startswith("s_setreg_imm32"):
            return 8
        return 8 if has_literal(line) else 4
    return 8


def assemble(clang, src, obj):
    r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj],
                       capture_output=True, text=True)
    return r.returncode, r.stderr


def verify(objdump, obj):
    """-> (number of 8-byte instructions, number of those at offset 4, {function: misaligned})"""
    out = subprocess.run([objdump, "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    addr_re = re.compile(r"//\s*([0-9A-Fa-f]+):")
    sym_re = re.compile(r"^[0-9a-f]+ <(.*)>:")
    rows = []
    cur = ""
    for l in out.splitlines():
        m = sym_re.match(l)
        if m:
            cur = m.group(1)
            continue
        m = addr_re.search(l)
        if m and l.startswith("\t"):
            rows.append((int(m.group(1), 16), cur, l.split()[0]))
    tot = bad = 0
    per = {}
    for (a, f, mn), (b, _, _) in zip(rows, rows[1:]):
        if b - a == 8 and mn != "s_nop" and mn != "s_code_end":
            tot += 1
            if a % 8:
                bad += 1
                per[f] = per.get(f, 0) + 1
    return tot, bad, per


def check_model(objdump, obj, text_lines):
    """Size model against the disassembly, function by function; reports mnemonics the model gets wrong."""
    outp = subprocess.run([objdump, "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    fn, dis = "", {}
    for l in outp.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            fn = m.group(1)
            dis[fn] = []
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
        if m and l.startswith("\t"):
            dis[fn].append(int(m.group(1), 16))
    wrong = {}
    cur, txt = None, []
    def flush():
        if cur in dis and len(dis[cur]) == len(txt):
            ad = dis[cur]
            for (l, mn), sz in zip(txt, [b - a_ for a_, b in zip(ad, ad[1:])]):
                if sz in (4, 8) and size_of(l, mn) != sz:
                    wrong[mn] = wrong.get(mn, 0) + 1
    for l in text_lines:
        m = re.match(r"^([A-Za-z_$][\w$.]*):", l)
        if m and not l.startswith(".L"):
            flush()
            cur, txt = m.group(1), []
            continue
        mn = is_instr(l)
        if mn:
            txt.append((l, mn))
    flush()
    if wrong:
        print("align_isa: size model wrong for", sorted(wrong.items(), key=lambda kv: -kv[1])[:12])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--clang", default="/opt/rocm/lib/llvm/bin/clang")
    ap.add_argument("--objdump", default="/opt/rocm/lib/llvm/bin/llvm-objdump")
    a = ap.parse_args()
    lines = open(a.src).read().split("\n")
    tmp = tempfile.mkdtemp(prefix="align_isa_")
    s1, o1 = os.path.join(tmp, "a.s"), os.path.join(tmp, "a.o")

    rc, err = assemble(a.clang, a.src, o1)
    if rc:
        sys.exit("align_isa: the input does not assemble:\n" + err[:2000])
    before = verify(a.objdump, o1)

    # 0. drop the conservative wait state behind our own single-instruction asm multiply-adds
    stripped = 0
    keep = []
    i = 0
    while i < len(lines):
        keep.append(lines[i])
        if (lines[i].strip() == ";;#ASMEND" and i >= 2 and lines[i - 2].strip() == ";;#ASMSTART" and is_instr(lines[i - 1]) == "v_mad_i64_i32"
                and i + 1 < len(lines) and lines[i + 1].strip() == "s_nop 0"):
            i += 2
            stripped += 1
            continue
        i += 1
    lines = keep

    # (Round 4 tried to shrink the callee-entry wait of the shared multiplier bodies - `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)`, which every
    # non-kernel function opens with - to lgkmcnt(0): they touch no vector memory.  UNSAFE: hipcc's callers rely on that wait.  They load
    # straight into the argument registers and call without waiting (fp_pow_sched: scratch_load_dwordx4 v[14:17] ... s_swappc_b64), so
    # the callee read stale arguments and k_hash_map never ended.  Not done.)
    relaxed = 0

    # 1. _e32 -> _e64 wherever the assembler takes it
    conv = {}
    for i, l in enumerate(lines):
        mn = is_instr(l)
        if mn and mn.endswith("_e32") and not has_literal(l):
            conv[i] = l
            lines[i] = l.replace(mn, mn[:-4] + "_e64", 1)
    for _ in range(20):
        open(s1, "w").write("\n".join(lines))
        rc, err = assemble(a.clang, s1, o1)
        if rc == 0:
            break
        bad = {int(m.group(1)) - 1 for m in re.finditer(r"a\.s:(\d+):\d+: error", err)}
        if not bad:
            sys.exit("align_isa: assembler error not tied to a line:\n" + err[:2000])
        for i in bad:
            if i in conv:
                lines[i] = conv.pop(i)
            else:
                sys.exit("align_isa: unexpected assembler error at line %d:\n%s" % (i + 1, err[:2000]))
    else:
        sys.exit("align_isa: conversion did not converge")

    # 2 + 3. 8-byte function alignment, s_nop in front of 8-byte instructions that would sit at offset 4
    out = []
    off = 0                 # offset modulo 8 inside the current aligned region
    pending = []            # labels / directives since the last instruction (a nop goes in front of them)
    nops = 0
    locked = 0              # instructions left in a PC-relative group (after s_getpc_b64)
    for l in lines:
        st = l.strip()
        if st.startswith(".p2align"):
            n = int(re.match(r"\.p2align[lw]?\s+(\d+)", st).group(1))
            if n < 3:
                l = l.replace(".p2align\t%d" % n, ".p2align\t3").replace(".p2align %d" % n, ".p2align 3")
            out.extend(pending)
            pending = []
            out.append(l)
            off = 0
            continue
        mn = is_instr(l)
        if mn is None:
            if st.startswith((".section", ".text", ".rodata", ".amdhsa_kernel", ".amdgpu_metadata")):
                out.extend(pending)
                pending = []
                out.append(l)
                continue
            pending.append(l)
            continue
        sz = size_of(l, mn)
        if mn == "s_getpc_b64":
            # s_getpc_b64 returns the address of the NEXT instruction and the s_add_u32 / s_addc_u32 (or the
            # .Lpost_getpc label) that follow are written relative to it: nothing may be inserted inside that
            # group.  Put the 4-byte s_getpc itself at offset 4 so that the group behind it starts aligned.
            if off == 0:
                out.append("\ts_nop 0")
                nops += 1
                off = 4
            locked = 2
        elif sz == 8 and off == 4:
            if locked > 0:
                pass                                 # inside a PC-relative group: leave it
            elif any(".Lpost_getpc" in x for x in pending):
                out.extend(pending)                  # keep the label glued to s_getpc_b64: pad after it
                pending = []
                out.append("\ts_nop 0")
                nops += 1
                off = 0
            else:
                out.append("\ts_nop 0")
                nops += 1
                off = 0
        if mn != "s_getpc_b64" and locked > 0:
            locked -= 1
        out.extend(pending)
        pending = []
        out.append(l)
        off = (off + sz) % 8
    out.extend(pending)
    open(a.dst, "w").write("\n".join(out))
    rc, err = assemble(a.clang, a.dst, o1)
    if rc:
        sys.exit("align_isa: the output does not assemble:\n" + err[:2000])
    after = verify(a.objdump, o1)
    check_model(a.objdump, o1, out)
    worst = sorted(after[2].items(), key=lambda kv: -kv[1])[:5]
    print("align_isa: %d of %d 8-byte instructions misaligned before, %d of %d after; %d _e32 re-encoded, %d s_nop inserted, %d asm wait states removed, %d callee-entry waits relaxed; worst: %s"
          % (before[1], before[0], after[1], after[0], len(conv), nops, stripped, relaxed, worst))


if __name__ == "__main__":
    main()
