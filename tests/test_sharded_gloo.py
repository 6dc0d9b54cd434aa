"""N > 1 path on CPU: two processes over gloo run the package's sharded driver with a stand-in cache
whose shard/merge arithmetic is the ORACLE (checker), so the partition plan, blinding-chain ids,
the 584-byte all_gather exchange and the merge/verdict logic are exercised without a GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleCache:
    """Same duck type as BatchedBLSVerifierCache.shard_device / finalverify_shards, oracle arithmetic."""

    def __init__(self, records, numThreads):
        self.records = records          # the GLOBAL host buffer; local_sets_ptr is a byte offset into it
        self.numThreads = numThreads

    def shard_device(self, ptr, n_total, lo, hi, rnd, stream=0):
        import bls12381_py as o
        from util import fp12_to_bytes
        chunks = o.parallel_chunks(min(n_total, self.numThreads), n_total)
        pairs, agg, ok = [], None, True
        base = chunks[lo][0]
        assert ptr == 320 * base
        for c in range(lo, hi):
            off, ln = chunks[c]
            seed = o.blinding_seed(rnd, c)
            for i in range(off, off + ln):
                seed, r = o.blinding_next(seed)
                rec = self.records[320 * i:320 * i + 320]
                pk, msg, sig = o.g1_from_blst_affine(rec[:96]), rec[96:128], o.g2_from_blst_affine(rec[128:])
                if pk is None:
                    ok = False
                    continue
                agg = o.g2_add(agg, o.g2_mul(sig, r))
                pairs.append((o.g1_mul(pk, r), o.hash_to_g2(msg)))
        pairs.append((o.g1_neg(o.G1_GEN), agg))
        return fp12_to_bytes(o.miller_loop(pairs)), ok

    def finalverify_shards(self, states):
        import bls12381_py as o
        from util import fp12_from_bytes
        f = o.F12_ONE
        for s in states:
            f = o.f12mul(f, fp12_from_bytes(s))
        return o.final_exp(f) == o.F12_ONE


def _worker(rank, world, port, case_name, tamper, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    from util import golden
    pkg = ge.load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("nim_blscurve_amd.sharded", os.path.join(ROOT, "nim-blscurve_amd", "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    c = [x for x in golden("batch")["cases"] if x["name"] == case_name][0]
    rec = bytearray(bytes.fromhex(c["sets"]))
    if tamper:
        rec[320 * (c["n"] - 1) + 100] ^= 1
    rec, rnd, n = bytes(rec), bytes.fromhex(c["rnd"]), c["n"]
    cache = OracleCache(rec, numThreads=4)

    def all_gather(blob):
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [bytes(x.numpy().tobytes()) for x in out]

    lo, hi, first, count = sh.shard_plan(n, 4, world)[rank]
    verdict = sh.batch_verify_sharded(cache, 320 * first, n, rank, world, rnd, all_gather)
    if rank == 0:
        q.put(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case_name,tamper,expect", [("n9", False, True), ("n9", True, False), ("inf_pk", False, False)])
def test_two_process_gloo(case_name, tamper, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case_name, tamper, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is expect


def test_shard_plan_partitions():
    sys.path.insert(0, ROOT)
    import importlib.util
    import bls12381_py as o
    spec = importlib.util.spec_from_file_location("sharded", os.path.join(ROOT, "nim-blscurve_amd", "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for n, t, w in [(17, 4, 2), (1 << 20, 8 * 4096, 8), (5, 4096, 8), (3, 4, 8), (65536 * 2, 8192, 2), (1000, 7, 3)]:
        plan = sh.shard_plan(n, t, w)
        chunks = o.parallel_chunks(min(n, t), n) if n < 10 ** 5 else None
        assert plan[0][0] == 0 and plan[-1][1] == min(n, t)
        nxt_c, nxt_t = 0, 0
        for lo, hi, first, count in plan:
            assert lo == nxt_c and first == nxt_t
            if chunks is not None and hi > lo:
                assert first == chunks[lo][0] and count == sum(c[1] for c in chunks[lo:hi])
            nxt_c, nxt_t = hi, first + count
        assert nxt_t == n


def _msm_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    import bls12381_py as o
    from util import golden, g1_aff_to_jac_bytes, g1_jac_to_affine
    spec = importlib.util.spec_from_file_location("nim_blscurve_amd.sharded", os.path.join(ROOT, "nim-blscurve_amd", "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    v = [x for x in golden("msm")["msm"] if x["n"] >= n][0]
    pts, sc = bytes.fromhex(v["points"])[:96 * n], bytes.fromhex(v["scalars"])[:32 * n]
    P = [o.g1_from_blst_affine(pts[96 * i:96 * i + 96]) for i in range(n)]
    K = [int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(n)]

    def local_partial(first, count):          # oracle stand-in for mi355_bls_p1s_mult_pippenger_partial_device
        return g1_aff_to_jac_bytes(o.msm_g1(P[first:first + count], K[first:first + count], 255))

    def add_partials(parts):                  # oracle stand-in for mi355_bls_p1s_add
        acc = None
        for b in parts:
            acc = o.g1_add(acc, g1_jac_to_affine(b))
        return g1_aff_to_jac_bytes(acc)

    def all_gather(blob):
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [bytes(x.numpy().tobytes()) for x in out]

    res = sh.msm_sharded(local_partial, add_partials, n, rank, world, all_gather)
    if rank == 0:
        q.put(g1_jac_to_affine(res) == o.msm_g1(P, K, 255))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 31])
def test_msm_two_process_gloo(n):
    """Point-sharded MSM protocol over gloo, world_size 2: shard ranges, one 144-byte all_gather, merge on rank 0 (oracle arithmetic
    standing in for the device entry points); n = 1 leaves rank 1 with an empty shard."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + n
    procs = [ctx.Process(target=_msm_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_msm_shard_range_matches_the_library():
    """The python mirror of the MSM shard plan == mi355_bls_msm_shard_range (a host function: no GPU needed)."""
    sys.path.insert(0, ROOT)
    import importlib.util
    import __graft_entry__ as ge
    m = ge.load_package()
    spec = importlib.util.spec_from_file_location("sharded", os.path.join(ROOT, "nim-blscurve_amd", "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for n, w in [(1 << 20, 8), (7, 8), (0, 3), (100, 3), ((1 << 17) + 5, 5)]:
        covered = 0
        for r in range(w):
            assert sh.msm_shard_range(n, w, r) == m.msm_shard_range(n, w, r)
            f, c = sh.msm_shard_range(n, w, r)
            assert f == covered
            covered += c
        assert covered == n
