#!/usr/bin/env python3
"""Static look at the loops of a gfx950 kernel / function in hipcc's -S output: for every natural loop (a backward branch to
an earlier label) the instruction mix of its body - VALU, 64-bit multiply-adds, scratch (spill) traffic, LDS, AGPR moves,
calls.  Scratch instructions inside a hot loop are register spills that go to HBM at 65 536 lanes; this is how they are
found without a GPU.   usage: tools/isa_loops.py dev.s <function-name-substring> [...]"""
import collections
import re
import sys


def functions(lines):
    out, cur, name = {}, None, None
    for l in lines:
        m = re.match(r'^([A-Za-z_][\w.$]*):', l)
        if m and not l.startswith('.L'):
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if l.startswith('.Lfunc_end'):
            cur = None
            continue
        if cur is not None:
            cur.append(l)
    return out


def mix(body):
    c = collections.Counter()
    for l in body:
        s = l.strip()
        if not l.startswith('\t') or not s or s[0] in '.;':
            continue
        op = s.split()[0]
        c['ins'] += 1
        if op.startswith('v_'):
            c['valu'] += 1
        if op in ('v_mad_i64_i32', 'v_mad_u64_u32'):
            c['mad'] += 1
        if op.startswith('scratch_'):
            c['scratch'] += 1
            w = {'dword': 1, 'dwordx2': 2, 'dwordx3': 3, 'dwordx4': 4}.get(op.split('_')[-1], 1)
            c['scratch_st_words' if 'store' in op else 'scratch_ld_words'] += w
        if op.startswith('ds_'):
            c['lds'] += 1
        if op.startswith('v_accvgpr'):
            c['agpr_mov'] += 1
        if op == 's_swappc_b64':
            c['calls'] += 1
    return c


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, l in enumerate(body):
        s = l.strip()
        m = re.match(r'^s_c?branch\w*\s+(\.LBB\w+)', s)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            res.append((labels[m.group(1)], i, m.group(1)))
    # merge loops with the same header (keep the widest)
    best = {}
    for a, b, n in res:
        if n not in best or b > best[n][1]:
            best[n] = (a, b, n)
    return sorted(best.values())


def main():
    lines = open(sys.argv[1]).read().split('\n')
    fns = functions(lines)
    for pat in sys.argv[2:]:
        for name, body in fns.items():
            if pat not in name:
                continue
            c = mix(body)
            print("%s\n  whole: %s" % (name[:100], dict(c)))
            ls = loops(body)
            for a, b, n in ls:
                depth = sum(1 for x, y, _ in ls if x <= a and b <= y) - 1
                print("  %sloop %s [%d..%d]: %s" % ("  " * depth, n, a, b, dict(mix(body[a:b + 1]))))


if __name__ == "__main__":
    main()
