"""The generated hand-allocated assembly loops built on nim-blscurve_amd/tools/asmlib.py (round 5): asmlib's one-lane interpreter executes
the SAME instruction tuples that become the kernel's text and checks every step against big-integer arithmetic - coordinates of the walking
point, the three line coefficients, the declared limb and value bounds of every stored value - with the field operations' preconditions
(fp.hpp's worst-case bookkeeping) asserted while generating.  No GPU needed; the GPU parity tests cover the assembled kernels."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOOLS = os.path.join(ROOT, "nim-blscurve_amd", "tools")


def run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(TOOLS, script)] + list(args), capture_output=True, text=True, cwd=TOOLS)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_lines_loop_whole_walk_matches_bigint_model():
    """k_lines: single doubling / addition steps and the whole 68-step walk (63 doublings, 5 additions, 204 line coefficients)."""
    out = run("gen_lines_asm.py", "--selftest")
    assert "selftest ok" in out
    m = re.search(r"doubling step (\d+) VALU instructions \((\d+) multiply-adds", out)
    valu, mads = int(m.group(1)), int(m.group(2))
    # 9 800 operand / reduction multiply-adds of the step's 22 Montgomery reductions (pairing.hpp's census) + 60 of the two scaled partial reductions
    assert mads == 9860
    assert valu <= 13100, "the doubling step grew: %d instructions" % valu


def test_lines_text_is_one_statement():
    t = run("gen_lines_asm.py")
    assert "#define BLS_LINES_ASM_BODY" in t and "#define BLS_LINES_ASM_CLOBBERS" in t
    assert "scratch_" not in t and "s_waitcnt vmcnt" not in t      # no spills; the line stores are never waited for
    assert t.count("s_setpc_b64") == 4                             # four multiplier subroutines
    assert '"v249"' in t and '"v250"' not in t and '"a255"' in t
    # every store of a line coefficient goes through the running row pointer with the lane's 32-bit byte offset
    assert t.count("global_store_dwordx4") == 2 * 3 * 2 * 3 and t.count("global_store_dwordx2") == 2 * 3 * 2


def test_clear_cofactor_kernel_body_matches_bigint_model():
    """k_hash_clear: single doubling / addition / psi steps with their bounds, the flag on crafted exceptional inputs, and the WHOLE kernel
    body (loads, P = q0 + q1, both chains, psi maps, the seven outer additions, the store - memory operations emulated) against the
    reference formula as a group element."""
    out = run("gen_clear_asm.py", "--selftest")
    assert "selftest ok" in out
    m = re.search(r"whole kernel body: (\d+) VALU instructions, (\d+) multiply-adds", out)
    valu, mads = int(m.group(1)), int(m.group(2))
    import bench
    assert abs(mads - bench.MAD_PER_TUPLE["k_hash_clear"]) <= 0.02 * bench.MAD_PER_TUPLE["k_hash_clear"]      # the census bench.py's roofline uses
    assert valu <= 1_370_000, "the kernel body grew: %d instructions" % valu
    assert mads / valu > 0.785


def test_clear_text_is_one_statement():
    t = run("gen_clear_asm.py")
    assert "#define BLS_CLEAR_ASM_BODY" in t and "#define BLS_CLEAR_ASM_CLOBBERS" in t
    assert "scratch_" not in t
    assert t.count("s_setpc_b64") == 8                             # four leaf multipliers, PREP, ADD, PSI, CHAIN
    assert '"v249"' in t and '"v250"' not in t and '"a255"' in t


def test_asmlib_reduce_and_carry_against_bigints():
    """the two normalisation steps of asmlib.Builder on extreme inputs: fp_carry_step and fp_reduce with a folded scale"""
    sys.path.insert(0, TOOLS)
    import random
    import asmlib as al
    a = al.Asm(100, 36, 50, 51, 52)
    b = al.Builder(a, list(range(84, 98)), 110, 111)
    x, d = al.blk(0), al.blk(14)
    rnd = random.Random(5)
    for scale in (1, 3, 12, -16):
        for trial in range(40):
            a.ins = []
            vb = 1024 // abs(scale) if trial % 2 else 2
            val = rnd.randrange(-vb * al.P, vb * al.P)
            # limbs: canonical ones shifted around by up to the allowed units (carry-less sums)
            ls = al.limbs_of(val % (1 << 392)) if val >= 0 else None
            if ls is None:
                ls = [(-l) & 0xffffffff for l in al.limbs_of((-val) % (1 << 392))]
            mach = al.Machine(a)
            al.put(mach, x, ls)
            assert al.get(mach, x) == val
            r = b.reduce(d, x.like(vb, 1), scale)
            mach.run(a.ins)
            got = al.get(mach, d)
            assert (got - scale * val) % al.P == 0 and abs(got) < 0.51 * al.P, (scale, val, got)
            al.check_limbs(mach, d, 0)
    for trial in range(40):
        a.ins = []
        ls = [rnd.randrange(-(7 << 28), 7 << 28) & 0xffffffff for _ in range(al.NL)]
        mach = al.Machine(a)
        al.put(mach, x, ls)
        val = al.get(mach, x)
        b.carry(x, x.like(4, 7))
        mach.run(a.ins)
        assert al.get(mach, x) == val
        al.check_limbs(mach, x, 1)


def test_msm_bucket_loop_matches_bigint_model():
    """k_pip_bucket<fp>: the mixed addition in extended Jacobian coordinates on random points with both signs, the final conversion, and the
    complete formula's exceptional branches driven through the loop's own control flow (q + q: the doubling path; q - q: infinity; infinity + r)."""
    out = run("gen_msm_asm.py", "--selftest")
    assert "selftest ok" in out
    m = re.search(r"(\d+) VALU instructions per addition \((\d+) multiply-adds", out)
    valu, mads = int(m.group(1)), int(m.group(2))
    assert mads == 3600 and valu <= 4500          # 6 products, 2 squares, one two-term dot product (3 542) + the two partial reductions (58)


def test_msm_text_is_one_statement():
    t = run("gen_msm_asm.py")
    assert "#define BLS_MSM_ASM_BODY" in t and "#define BLS_MSM_ASM_CLOBBERS" in t
    assert "scratch_" not in t and "s_swappc" not in t            # no spills, no calls: the multiplier bodies are expanded in place
    assert '"v229"' in t and '"v230"' not in t and '"a0"' not in t  # 230 VGPRs, no AGPRs: two waves per SIMD


def test_pkmul_blocks_match_bigint_model():
    """k_pkmul: the table of 1 .. 8 times the key with each entry's Z^2, Z^3, then 17 windows of four doublings and one signed table addition
    (biased digits), block by block against big-integer Jacobian arithmetic; acc == entry is detected (Z3 == 0)."""
    out = run("gen_pkmul_asm.py", "--selftest")
    assert "selftest ok" in out
    m = re.search(r"doubling (\d+) instructions \((\d+) multiply-adds\), addition (\d+) \((\d+)\)", out)
    nd, md, na, ma = (int(x) for x in m.groups())
    assert md == 2334 and ma == 5077 and nd <= 3050 and na <= 6700


def test_pkmul_text_is_one_statement():
    t = run("gen_pkmul_asm.py")
    assert "#define BLS_PKMUL_ASM_BODY" in t and "#define BLS_PKMUL_ASM_CLOBBERS" in t
    assert "scratch_" not in t and '"a0"' not in t and '"v241"' in t and '"v242"' not in t      # VGPRs only, two waves per SIMD
    assert t.count("v_mad_i64_i32") < 7000          # the hot code (two shared bodies, the doubling, the addition's glue) stays near 40 KB


def test_pow_chain_matches_python_pow():
    """fp_recip_sqrt_pow's device body (round 6, tools/gen_pow_asm.py): a^((p-3)/4) through the whole unrolled 4-bit sliding-window schedule against
    Python's pow, on a canonical residue, on 1, and on the widest input the contract allows (a limb-wise sum of two residues)."""
    out = run("gen_pow_asm.py", "--selftest")
    assert "selftest ok" in out
    m = re.search(r"(\d+) squarings \+ (\d+) window products \+ 8 for the table; (\d+) VALU instructions, (\d+) multiply-adds", out)
    nsq, nmul, valu, mads = map(int, m.groups())
    assert nsq + 1 + 3 == 379 and nmul + 8 <= 86          # 379 doublings of the exponent in all (one in the table, the top window holds three bits)
    assert mads / valu > 0.79
    t = run("gen_pow_asm.py")
    assert "#define BLS_POW_ASM_BODY" in t and "scratch_" not in t and "accvgpr" not in t       # the table in VGPRs: no AGPR, no memory
    assert '"v171"' in t and '"v172"' not in t
