#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from the built assembly (amdhsa.kernels metadata of nim-blscurve_amd/build/dev_aligned.s).
usage: python3 tools/kernel_metadata.py [dev_aligned.s] > profiles/r0N_kernel_metadata.txt
       python3 tools/kernel_metadata.py --json [dev_aligned.s] > nim-blscurve_amd/libblscurve_mi355x.so.kmeta.json   (build.sh does this: the table travels
       with the library, so that bench.py can put the spill counts of the kernels it ran into its line on a box that has no build/ directory)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
as_json = "--json" in sys.argv
args = [x for x in sys.argv[1:] if x != "--json"]
t = open(args[0] if args else os.path.join(ROOT, "nim-blscurve_amd", "build", "dev_aligned.s")).read()
blocks = t[t.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]
rows = []
for b in blocks:
    b = ".agpr_count:" + b
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    rows.append([g(k) for k in ("name", "vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")])
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
def short(d):
    n = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)
if as_json:
    def num(x):
        return int(x) if x.isdigit() else None
    print(json.dumps({short(d): {"vgpr": num(r[1]), "agpr": num(r[2]), "spill": num(r[3]), "scratch": num(r[4]), "lds": num(r[5]), "wg": num(r[6])}
                      for r, d in zip(rows, dem)}, indent=0, sort_keys=True))
    sys.exit(0)
print("# Per-kernel metadata of the built library (build/dev_aligned.s, amdhsa.kernels).  vgpr = arch VGPRs + AGPRs allocated (unified file, 512 per lane);")
print("# spill = vgpr_spill_count; scratch = private_segment_fixed_size, bytes per lane (includes stack objects of out-of-line callees such as fp_inv);")
print("# lds = group_segment_fixed_size, bytes per workgroup; wg = max_flat_workgroup_size.")
print("%-48s %5s %5s %6s %8s %7s %5s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "lds", "wg"))
for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
    n = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    print("%-48s %5s %5s %6s %8s %7s %5s" % (n[:48], r[1], r[2], r[3], r[4], r[5], r[6]))
