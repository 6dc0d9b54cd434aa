#!/usr/bin/env python3
"""Per-loop instruction mix of a function in hipcc's -S output from the compiler's own block comments ("Loop Header", "in Loop: Header=BBx_y Depth=d"):
unlike isa_loops.py this also sees loops closed by a long jump (s_getpc / s_setpc).  usage: tools/isa_blocks.py dev.s <function-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
cur = None; inside = False
blocks = []   # (label, header-of-innermost-loop or None, depth, counter)
for l in lines:
    m = re.match(r'^([A-Za-z_][\w.$]*):', l)
    if m and not l.startswith('.L'):
        inside = pat in m.group(1)
        if inside: print(m.group(1)[:120])
        continue
    if not inside: continue
    if l.startswith('.Lfunc_end'): inside = False; continue
    m = re.match(r'^(\.LBB\w+):\s*;?(.*)', l)
    if m:
        lab, com = m.group(1), m.group(2)
        hdr, depth = None, 0
        mm = re.search(r'Header=(BB\w+) Depth=(\d+)', com)
        if mm: hdr, depth = '.L' + mm.group(1), int(mm.group(2))
        mm = re.search(r'Loop Header: Depth=(\d+)', com)
        if mm: hdr, depth = lab, int(mm.group(1))
        cur = [lab, hdr, depth, collections.Counter()]
        blocks.append(cur)
        continue
    if cur is None:
        cur = ['entry', None, 0, collections.Counter()]; blocks.append(cur)
    s = l.strip()
    if not l.startswith('\t') or not s or s[0] in '.;': 
        # continuation comment lines of a label ("Parent Loop ...") refine header info
        mm = re.search(r';\s+Parent Loop (BB\w+) Depth=(\d+)', l)
        continue
    op = s.split()[0]
    c = cur[3]
    c['ins'] += 1
    if op.startswith('v_'): c['valu'] += 1
    if op in ('v_mad_i64_i32', 'v_mad_u64_u32'): c['mad'] += 1
    if op.startswith('scratch_'): c['scratch'] += 1
    if op.startswith('ds_'): c['lds'] += 1
    if op.startswith(('global_', 'flat_')): c['vmem'] += 1
    if op.startswith('v_accvgpr'): c['agpr_mov'] += 1
    if op.startswith('v_mov_b32'): c['v_mov'] += 1
    if op == 's_swappc_b64': c['calls'] += 1
    if op.startswith('s_waitcnt'): c['waitcnt'] += 1
agg = collections.OrderedDict()
for lab, hdr, depth, c in blocks:
    key = (hdr, depth)
    agg.setdefault(key, collections.Counter()).update(c)
for (hdr, depth), c in agg.items():
    print('  %s depth=%d %s' % (hdr, depth, dict(c)))
