#!/usr/bin/env python3
"""Writes tests/golden/cabi_fixture.bin from batch.json / msm.json (themselves made by gen_golden.py from the
KAT-pinned oracle): the inputs tests/cabi/cabi_caller.c feeds through the C ABI.  Layout (little-endian):
  "MI355CAB" | u32 n_sets | n_sets x 320 B SignatureSet records (golden case n17) | 32 B rnd
  | u32 n_bad | n_bad x 320 B (golden case forged_among_many) | 32 B rnd
  | u32 n_pts | u32 nbits | n_pts x 96 B blst_p1_affine | n_pts x 32 B scalars (golden MSM n = 32)"""
import json
import os
import struct

here = os.path.dirname(os.path.abspath(__file__))
b = json.load(open(os.path.join(here, "batch.json")))["cases"]
good = [c for c in b if c["name"] == "n17"][0]
bad = [c for c in b if c["name"] == "forged_among_many"][0]
msm = [v for v in json.load(open(os.path.join(here, "msm.json")))["msm"] if v["n"] == 32][0]
out = b"MI355CAB"
for c in (good, bad):
    out += struct.pack("<I", c["n"]) + bytes.fromhex(c["sets"]) + bytes.fromhex(c["rnd"])
out += struct.pack("<II", msm["n"], msm["nbits"]) + bytes.fromhex(msm["points"]) + bytes.fromhex(msm["scalars"])
open(os.path.join(here, "cabi_fixture.bin"), "wb").write(out)
print(len(out), "bytes")
