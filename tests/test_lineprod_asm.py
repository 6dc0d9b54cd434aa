"""The generator of k_lineprod's hand-allocated assembly loop (nim-blscurve_amd/tools/gen_lineprod_asm.py): its own one-lane interpreter
executes the generated instruction list and checks f * line against big-integer arithmetic (no GPU).  The GPU parity tests
(test_gpu_batch / test_gpu_headline: GT values against the C oracle) cover the assembled loop itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "nim-blscurve_amd", "tools", "gen_lineprod_asm.py")


def test_generated_line_product_matches_bigint_model():
    r = subprocess.run([sys.executable, GEN, "--selftest"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout
    assert "16464 multiply-adds" in r.stdout          # 12 dot products of six terms: the census bench.MAD_PER_TUPLE["k_lineprod"] / 68 lines


def test_emitted_text_is_one_statement_with_clobbers():
    r = subprocess.run([sys.executable, GEN], capture_output=True, text=True)
    assert r.returncode == 0
    t = r.stdout
    assert "#define BLS_LINEPROD_ASM_BODY" in t and "#define BLS_LINEPROD_ASM_CLOBBERS" in t
    assert t.count("v_mad_i64_i32") == 2744                       # ONE copy of the coefficient subroutine: two dot products of 6 x 196 + 196
    assert t.count("s_swappc_b64") == 6                           # called once per coefficient of the line product
    assert "scratch_" not in t                                    # no spills
    assert '"a255"' in t and '"v249"' in t and '"v250"' not in t  # v250.. stay the compiler's


def test_built_isa_has_no_dpp_folded_into_a_consumer():
    """gfx950 applies a DPP lane permutation on a "rev" VOP2 opcode (v_subrev_u32, v_lshlrev_b32 ...) to src1, while LLVM's DPP combine folds
    v_mov_b32_dpp into such consumers assuming src0 (measured: tools/test_dpp.hip; it made a - b wrong in the even lanes of the G1 lane teams).
    build.sh therefore passes -mllvm -amdgpu-dpp-combine=false: every DPP instruction of the built library must be a plain v_mov_b32_dpp."""
    isa = os.path.join(ROOT, "nim-blscurve_amd", "build", "dev_aligned.s")
    if not os.path.exists(isa):
        import __graft_entry__ as ge
        ge.build()
    assert "-amdgpu-dpp-combine=false" in open(os.path.join(ROOT, "nim-blscurve_amd", "build.sh")).read()
    dpp = [l.split()[0] for l in open(isa) if "_dpp" in l and not l.lstrip().startswith((";", "//", "."))]
    assert dpp, "the lane teams exchange with DPP quad permutes: none found"
    assert set(dpp) == {"v_mov_b32_dpp"}, sorted(set(dpp))
