"""The lane-team engine of the latency path (round 6; nim-blscurve_amd/csrc/teamvm.hpp, programs from nim-blscurve_amd/tools/teamvm.py), on the CPU:
  * the programs executed on Python integers - from the round objects and again from the ENCODED tables alone (what the kernel reads) - against
    big-integer formulas: the whole cofactor clearing of hash-to-G2 as a group element, all 68 x 6 line coefficients of the Miller walk;
  * the engine's own C++ (tvm_product / tvm_post: the arithmetic the device lanes run) on sixteen emulated lanes under the bounds tracker
    (tests/host_emu), against the product's one-lane formulas: clear_cofactor_g2 of the oracle, pairing.hpp's miller_lines.
The GPU parity tests (tests/test_gpu_clear_chain.py[latency], test_gpu_fav.py, test_gpu_batch.py small batches) run the kernels themselves.
Reference: blst_abi.nim:383 (hash-to-G2's cofactor clearing), :455 (miller_loop_n)."""
import os
import random
import re
import subprocess
import sys

import bls12381_py as o
from util import buf, g1_aff_to_jac_bytes, g2_jac_to_affine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "nim-blscurve_amd", "tools")


def test_programs_match_bigint_formulas():
    r = subprocess.run([sys.executable, os.path.join(TOOLS, "teamvm.py"), "--selftest"], capture_output=True, text=True, cwd=TOOLS)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "teamvm selftest ok" in r.stdout
    m = re.search(r"teamvm clear: (\d+) slots, (\d+) distinct rounds, (\d+) in sequence", r.stdout)
    slots, distinct, nseq = map(int, m.groups())
    # two chains of 63 doublings (3 rounds) + 5 additions (5 rounds), 8 more additions, 9 preparations (2 rounds), one doubling, three psi, copies
    assert nseq <= 490 and distinct <= 40
    assert 4 * slots * 80 <= 25 * 1024            # four teams of a wave at the 80-byte slot stride: six waves per CU fit the 160 KB of LDS
    m = re.search(r"teamvm lines: (\d+) slots, (\d+) distinct rounds, (\d+) in sequence \((\d+) linear\)", r.stdout)
    slots, distinct, nseq, nlin = map(int, m.groups())
    assert nseq == 2 + 63 * 4 + 5 * 5 and nlin == 63
    assert 4 * slots * 80 <= 27 * 1024


def test_tables_are_well_formed():
    sys.path.insert(0, TOOLS)
    try:
        import teamvm as tv
    finally:
        sys.path.pop(0)
    for obj in (tv.G2Clear(), tv.Lines()):
        prog = obj.build()
        words = tv.encode(prog)
        assert len(words) == len(prog.rounds) * 16 * 4
        for i in range(0, len(words), 4):
            w0, w1, w2, w3 = words[i:i + 4]
            for off in (w0 & 0xffff, w0 >> 16, w1 & 0xffff, w1 >> 16):
                assert off % tv.SLOT_BYTES == 0 and off // tv.SLOT_BYTES < obj.s.n               # every operand / destination is a slot of the team's region
            assert (w1 >> 16) // tv.SLOT_BYTES != prog.zero                            # nothing ever writes the zero slot
            cs = [(w2 >> (8 * k)) & 0xff for k in range(4)] + [w3 & 0xff]
            assert sum(c - 256 if c & 0x80 else c if c < 0x80 else 0 for c in cs) is not None
            assert sum(abs(c - 256 if c & 0x80 else c) for c in cs) <= 64   # the reduction's quotient estimate
        for e in prog.seq:
            assert (e & 0xffff) < len(prog.rounds)
            if e & tv.F_GSTORE:
                assert ((e >> tv.STEP_SHIFT) & 0xff) < 68


def _jac_bytes(p, z):
    z2 = o.f2sqr(z)
    x, y = o.f2mul(p[0], z2), o.f2mul(p[1], o.f2mul(z2, z))
    return b"".join(o.fp_to_mont_bytes(c) for c in (x[0], x[1], y[0], y[1], z[0], z[1]))


import pytest


@pytest.fixture(params=["tvm", "rvm"])
def engine(request, emu):
    """the same programs on the lane-team executor (teamvm.hpp: a lane per product) and on the row executor (rowvm.hpp: a DPP row per product)"""
    class E:
        clear = getattr(emu, "emu_%s_clear" % request.param)
        lines_equal = getattr(emu, "emu_%s_lines_equal" % request.param)
    return E


def test_engine_clears_cofactor_like_the_oracle(engine):
    """tvm_run_host / rvm_run_host (bounds tracker on: every product's and every reduction's preconditions are asserted) on random pairs of E2 points"""
    emu = engine
    rng = random.Random(17)
    for _ in range(3):
        pts = [o.iso3_g2(o.sswu_g2((rng.randrange(o.P), rng.randrange(o.P)))) for _ in range(2)]
        zs = [(rng.randrange(1, o.P), rng.randrange(o.P)) for _ in range(2)]
        out = buf(288)
        emu.clear(_jac_bytes(pts[0], zs[0]) + _jac_bytes(pts[1], zs[1]), out)
        got = g2_jac_to_affine(out.raw)
        assert got == o.clear_cofactor_g2(o.g2_add(pts[0], pts[1]))
        assert o.g2_in_subgroup(got)


def test_engine_exceptional_additions_end_in_z_zero(engine):
    """q0 == q1, q0 == -q1, an operand at infinity: the incomplete additions must leave Z = 0 (k_clear_fix / k_hash_one then recompute)"""
    emu = engine
    rng = random.Random(19)
    a = o.iso3_g2(o.sswu_g2((rng.randrange(o.P), rng.randrange(o.P))))
    z = (rng.randrange(1, o.P), rng.randrange(o.P))
    for q0, q1 in ((_jac_bytes(a, z), _jac_bytes(a, (5, 7))), (_jac_bytes(a, z), _jac_bytes(o.g2_neg(a), (3, 1))), (bytes(288), _jac_bytes(a, z)),
                   (_jac_bytes(a, z), bytes(288))):
        out = buf(288)
        emu.clear(q0 + q1, out)
        assert out.raw[192:288] == bytes(96)


def test_engine_lines_equal_miller_lines(engine):
    emu = engine
    rng = random.Random(23)
    for _ in range(2):
        p = o.g1_mul(o.G1_GEN, rng.randrange(1, o.R))
        q = o.g2_mul(o.G2_GEN, rng.randrange(1, o.R))
        zp = rng.randrange(1, o.P)
        pj = b"".join(o.fp_to_mont_bytes(c) for c in (p[0] * zp * zp % o.P, p[1] * zp * zp * zp % o.P, zp))
        assert emu.lines_equal(pj, _jac_bytes(q, (rng.randrange(1, o.P), rng.randrange(o.P)))) == 1
    assert emu.lines_equal(g1_aff_to_jac_bytes(o.G1_GEN), _jac_bytes(o.G2_GEN, (1, 0))) == 1
