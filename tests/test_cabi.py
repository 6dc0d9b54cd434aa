"""CPU-only checks of the drop-in boundary: the library builds for gfx950, loads without a GPU and
exports every symbol include/blscurve_mi355x.h declares; host logic (packing, dispatch rules)."""
import ctypes
import re

import pytest


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def test_exports_match_header(m):
    hdr = open(m.HEADER_PATH).read()
    names = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 40 and "mi355_p1s_mult_pippenger" in names
    L = ctypes.CDLL(m.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_chunk_range_matches_parallel_chunks(m):
    import bls12381_py as o
    for n, t in [(17, 4), (2, 4), (100, 7), (4096, 4096), (5000, 4096), (1, 4)]:
        chunks = o.parallel_chunks(min(n, t), n)
        offs = [c[0] for c in chunks] + [n]
        B = len(chunks)
        picks = sorted({0, 1, B // 3, B // 2, B - 1, B} & set(range(B + 1)))
        for lo in picks:
            for hi in picks:
                if lo < hi:
                    first, count = m.chunk_range(n, t, lo, hi)
                    assert first == offs[lo] and count == offs[hi] - offs[lo]


def test_pack_and_validation(m):
    rec = m.pack_signature_sets([(bytes(96), bytes(32), bytes(192))] * 3)
    assert len(rec) == 960
    with pytest.raises(ValueError):
        m.pack_signature_sets([(bytes(95), bytes(32), bytes(192))])


def test_no_gpu_fails_loudly(m):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(m.BlsGpuError):
        m.BatchedBLSVerifierCache.init(max_sets=16)


def test_shard_plan_matches_python_plan(m):
    """mi355_bls_shard_plan (what the in-library multi-GPU driver uses) == sharded.shard_plan (what the bench uses)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("sharded", os.path.join(os.path.dirname(m.LIB_PATH), "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for n, t, w in [(17, 4, 2), (3, 4, 8), (1 << 20, 32768, 8), (1000, 64, 3), (5, 4096, 4), (1, 4, 2)]:
        assert [m.shard_plan(n, t, w, g) for g in range(w)] == sh.shard_plan(n, t, w)


def test_secure_random_bytes_must_be_32_bytes(m):
    for bad in (bytes(31), bytes(33), 32, "x" * 32):
        with pytest.raises(ValueError):
            m._rnd32(bad)
    assert m._rnd32(bytearray(32)) == bytes(32)
