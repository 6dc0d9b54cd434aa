#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (one directory per pass) into per-kernel, per-launch numbers.
usage: tools/summarize_pmc.py out.json dir1 [dir2 ...]
FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B requests as
64 B for wide coalesced streams (MI355X_MICROARCH.md, HBM section), so `hbm_bytes_corrected` doubles it."""
import collections
import csv
import glob
import json
import re
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from bench import KERNEL_BYTES, MAD_PER_TUPLE
except Exception:                      # torch missing: the shares are simply not computed
    KERNEL_BYTES, MAD_PER_TUPLE = {}, {}
TUPLES = 65536
# multiply-adds per LAUNCH of the Pippenger kernels at 2^20 points x 255 bits (tests/gpu_probe_aux.py msm): k_pip_bucket runs once per
# group of 8 windows, one mixed addition (8M + 2S = 3738 multiply-adds) per (point, window) with a non-zero digit
MSM_MAD_PER_LAUNCH = {"k_pip_bucket": (1 << 20) * 8 * (1 - 2.0 ** -16) * 3738}
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"::(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[m.group(1)][r["Counter_Name"]] += 1
res = {}
for k, v in agg.items():
    e = {c: x / calls[k][c] for c, x in v.items()}          # per launch
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        f, w = e.get("FETCH_SIZE", 0.0) * 1024, e.get("WRITE_SIZE", 0.0) * 1024
        e["hbm_bytes_raw"] = f + w
        e["hbm_bytes_corrected"] = 2 * f + w
    if "SQ_WAVE_CYCLES" in e and e.get("SQ_INSTS_VALU"):
        e["cycles_per_valu"] = 4 * e["SQ_WAVE_CYCLES"] / e["SQ_INSTS_VALU"]
    if e.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in e:
        e["wait_any_share"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_INSTS_VALU") and k in MSM_MAD_PER_LAUNCH:
        e["mad_share_of_valu"] = MSM_MAD_PER_LAUNCH[k] / 64.0 / e["SQ_INSTS_VALU"]
    if e.get("SQ_INSTS_VALU") and k in MAD_PER_TUPLE:
        # 64-bit multiply-adds (census of the formulas, bench.py MAD_PER_TUPLE) as a share of all VALU wave-instructions:
        # one wave-instruction serves 64 tuples (k_hash_map: two lanes per tuple, so 32)
        waves_per_tuple = (1 / 32.0) if k == "k_hash_map" else (1 / 64.0)
        e["mad_share_of_valu"] = MAD_PER_TUPLE[k] * TUPLES * waves_per_tuple / e["SQ_INSTS_VALU"] / (2 if k == "k_hash_map" else 1)
        e["algorithmic_bytes"] = KERNEL_BYTES.get(k, 0) * TUPLES
        if e.get("hbm_bytes_corrected") and e["algorithmic_bytes"]:
            e["hbm_over_algorithmic"] = e["hbm_bytes_corrected"] / e["algorithmic_bytes"]
    res[k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(res), "kernels")
