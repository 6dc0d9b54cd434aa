"""The drop-in boundary exercised by a compiled, non-Python caller: tests/cabi/cabi_caller.c (plain C, only
include/blscurve_mi355x.h) is built with gcc, linked against the in-tree library and run on the golden fixtures."""
import os
import subprocess

import pytest

import bls12381_py as o
from util import g1_jac_to_affine, golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    libdir = os.path.join(ROOT, "nim-blscurve_amd")
    exe = str(tmp_path / "cabi_caller")
    subprocess.check_call(["gcc", "-Wall", "-Wextra", "-std=c99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_caller.c"), "-o", exe, "-L", libdir,
                           "-l:libblscurve_mi355x.so", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "cabi_fixture.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    kv = {}
    for line in out.stdout.splitlines():
        k, v = line.rsplit(" ", 1)
        kv[k] = v
    # golden n17 verifies through every entry point, forged_among_many (t_batch_verifier.nim:198-244) through none
    assert kv["batch0 n"] == "17" and kv["batch1 n"] == "18"
    assert (kv["batch0 parallel"], kv["batch0 serial"], kv["batch0 once"]) == ("1", "1", "1")
    assert (kv["batch1 parallel"], kv["batch1 serial"], kv["batch1 once"]) == ("0", "0", "0")
    assert kv["empty"] == "0"
    assert kv["many"] == "0100"              # rc 0 (not all verified); golden n17 true, the empty batch false, forged_among_many false
    # a 5-set context (slices), two contexts (multi-device driver): the same verdicts
    assert (kv["batch0 sliced"], kv["batch0 sliced_serial"], kv["batch0 multi"]) == ("1", "1", "1")
    assert (kv["batch1 sliced"], kv["batch1 sliced_serial"], kv["batch1 multi"]) == ("0", "0", "0")
    # streaming aggregateVerify: every update accepted, a non-aggregate signature rejected; infinity key -> update and finish false
    for k in ("batch0", "batch1"):
        assert (kv[k + " aggv_updates"], kv[k + " aggv_finish"], kv[k + " aggv_inf_update"], kv[k + " aggv_inf_finish"]) == ("1", "0", "0", "0")
    # aggregateAll on signatures + finish(AggregateSignature): the naive aggregate check passes for BOTH batches (the forged pair of
    # t_batch_verifier.nim:198-244 is built to pass it), an aggregate that misses one signature fails
    for k in ("batch0", "batch1"):
        assert (kv[k + " g2_aggregate"], kv[k + " aggv_p2_finish"], kv[k + " aggv_p2_short"]) == ("0", "1", "0")
    want = [v for v in golden("msm")["msm"] if v["n"] == 32][0]["result_affine"]
    for k in ("msm_contiguous", "msm_pointer_list", "msm_ctx", "msm_multi", "msm_partials_added"):
        assert o.g1_to_blst_affine(g1_jac_to_affine(bytes.fromhex(kv[k]))).hex() == want, k
