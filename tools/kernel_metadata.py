#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from the built assembly (amdhsa.kernels metadata of nim-blscurve_amd/build/dev_aligned.s).
usage: python3 tools/kernel_metadata.py > profiles/r0N_kernel_metadata.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nim-blscurve_amd", "build", "dev_aligned.s")).read()
blocks = t[t.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]
rows = []
for b in blocks:
    b = ".agpr_count:" + b
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    rows.append([g(k) for k in ("name", "vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")])
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("# Per-kernel metadata of the built library (build/dev_aligned.s, amdhsa.kernels).  vgpr = arch VGPRs + AGPRs allocated (unified file, 512 per lane);")
print("# spill = vgpr_spill_count; scratch = private_segment_fixed_size, bytes per lane (includes stack objects of out-of-line callees such as fp_inv);")
print("# lds = group_segment_fixed_size, bytes per workgroup; wg = max_flat_workgroup_size.")
print("%-48s %5s %5s %6s %8s %7s %5s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "lds", "wg"))
for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
    n = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    print("%-48s %5s %5s %6s %8s %7s %5s" % (n[:48], r[1], r[2], r[3], r[4], r[5], r[6]))
