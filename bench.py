#!/usr/bin/env python3
"""Headline benchmark: BLS batch signature verifications / second on MI355X.

A step = one batchVerify of a 65 536-tuple batch per GPU (the size BASELINE.json's target is quoted
on), inputs resident in HBM before the timed region; `--inflight` (default 3) independent caller contexts keep
that many batches in flight so the serial tail of one overlaps the wide kernels of another.  N > 1: one process per GPU, each verifies its
own 65 536-tuple shard of one global batch (weak scaling); the only exchange is an all_gather of the
576-byte committed Fp12 state + ok flag per rank (RCCL), then one final exponentiation on rank 0.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HBM) and
`cpu_baseline` (the C restatement of the reference algorithm, oracle/bls_oracle.c, on the host cores).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic HBM bytes per tuple of each kernel, BLST-image sizes
KERNEL_BYTES = {"k_hash_map": 32 + 2 * 288, "k_hash_clear": 2 * 288 + 288, "k_pkmul": 96 + 8 + 144, "k_lines": 144 + 288 + 68 * 288,
                "k_lineprod": 68 * 288}
# which stage timer (HIP events inside the library) measures which single kernel
KERNEL_OF_STAGE = {"pk_mul": "k_pkmul", "miller_lines": "k_lines"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=65536, help="tuples per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=0, help="tuples in the CPU baseline sample (0 = auto)")
    ap.add_argument("--threads", action="store_true", help="one blocking call per host thread instead of submit / wait from one thread")
    ap.add_argument("--inflight", type=int, default=3, help="batches kept in flight per GPU (independent caller contexts)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the fastAggregateVerify / MSM side measurements")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DIST_BACKEND=gloo + BENCH_ALL_ON_DEVICE0=1 is a TEST HOOK: it lets the N > 1 code path (sharding, exchange,
    # merge) run end to end on a box with a single GPU; real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_ALL_ON_DEVICE0") == "1":
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    ge.build()
    m = ge.load_package()

    n = a.batch
    # caller streams first: HIP spreads streams over its hardware queues as they are created, and two callers
    # whose streams share a hardware queue run strictly one after the other (seen with rocprofv3 --kernel-trace)
    inflight = max(1, a.inflight)
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
    t0 = time.time()
    # n distinct valid (pk, SHA256("msg"+i), sig) tuples made by the library's own device signer
    # (mi355_bls_sign_sets_device; parity-tested against the oracle in tests/test_gpu_sign.py)
    gen = m.BatchedBLSVerifierCache.init(max_sets=n, device=local)
    d_sets = sign_records(m, gen, dev, range(rank * n, rank * n + n))
    del gen
    gen_s = time.time() - t0
    rnd = bytearray(hashlib.sha256(b"Mr F was here").digest())

    n_total = n * world
    nthreads = m.DEFAULT_NUM_THREADS * world                # global number of blinding chains
    # `inflight` independent callers (one context + stream each, "one context per concurrent caller",
    # bls_batch_verifier.nim:389-391) keep several batches in flight so that one batch's serial tail
    # (signature fold, Horner, final exponentiation: a handful of waves) overlaps another batch's wide kernels.
    caches = [m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nthreads, device=local) for _ in range(inflight)]
    cache = caches[0]
    fv_cache = m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=nthreads, device=local) if world > 1 else None

    import importlib.util
    from concurrent.futures import ThreadPoolExecutor
    spec = importlib.util.spec_from_file_location("nim_blscurve_amd.sharded", os.path.join(ROOT, "nim-blscurve_amd", "sharded.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    lo, hi, first, count = sharded.shard_plan(n_total, nthreads, world)[rank]
    assert count == n

    def all_gather(blob):
        mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        if backend == "nccl":
            mine = mine.to(dev)
        allst = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allst, mine)               # RCCL; 584 B per rank
        return [bytes(t.cpu().numpy().tobytes()) for t in allst]

    stage_acc = {}
    acc_lock = __import__("threading").Lock()
    free = __import__("queue").Queue()
    for i in range(inflight):
        free.put(i)

    def compute(it, record):
        """The per-batch GPU work of one step on a free caller context (a context is never shared by two
        calls at a time); world > 1: this rank's shard state."""
        slot = free.get()
        try:
            c, st = caches[slot], streams[slot].cuda_stream
            r = bytes(rnd)
            out = c.verify_device(d_sets.data_ptr(), n, r, st) if world == 1 else c.shard_device(d_sets.data_ptr(), n_total, lo, hi, r, st)
            if record:
                with acc_lock:
                    for k, v in list(c.timings().items()) + list(c.kernel_timings().items()):
                        stage_acc[k] = stage_acc.get(k, 0.0) + v
            return out
        finally:
            free.put(slot)

    def run_steps_async(k, record):
        """One host thread keeps `inflight` batches in flight with the submit / wait entry points (context
        i % inflight; a context is waited for right before it is reused, i.e. oldest first).  Every batch is
        chained to the one submitted before it (`after`): it starts when that one has finished hashing, so the batches
        in flight sit at different stages and a serial tail always runs beside whole-chip kernels of another batch.
        N > 1: the per-step all_gather + final exponentiation (rank 0) happen in step order as results arrive."""
        ok = True
        busy = [False] * inflight

        def collect(slot):
            c = caches[slot]
            busy[slot] = False
            if world == 1:
                res = c.wait()
            else:
                state, okf = c.shard_wait()
                blobs = all_gather(state + bytes([1 if okf else 0]) + bytes(7))
                res = True
                if rank == 0:
                    res = all(b[576] == 1 for b in blobs) and fv_cache.finalverify_shards([b[:576] for b in blobs])
            if record:
                for kk, v in list(c.timings().items()) + list(c.kernel_timings().items()):
                    stage_acc[kk] = stage_acc.get(kk, 0.0) + v
            return res

        r = bytes(rnd)
        for it in range(k):
            slot = it % inflight
            if busy[slot]:
                ok = collect(slot) and ok
            # chaining staggers whole-chip batches; small batches do not fill the chip and simply run side by side
            after = caches[(slot - 1) % inflight] if (inflight > 1 and n >= 32768) else None
            if world == 1:
                caches[slot].submit_device(d_sets.data_ptr(), n, r, streams[slot].cuda_stream, after=after)
            else:
                caches[slot].shard_submit_device(d_sets.data_ptr(), n_total, lo, hi, r, streams[slot].cuda_stream, after=after)
            busy[slot] = True
        for j in range(inflight):
            slot = (k + j) % inflight
            if busy[slot]:
                ok = collect(slot) and ok
        return ok

    def run_steps(k, record):
        """k steps; the collective and the verdict of every step are issued in step order on this thread."""
        if not a.threads:
            return run_steps_async(k, record)
        ok = True
        with ThreadPoolExecutor(max_workers=inflight) as pool:
            futs = [pool.submit(compute, i, record) for i in range(k)]
            for f in futs:
                res = f.result()
                if world == 1:
                    ok = ok and res
                else:
                    state, okf = res
                    blobs = all_gather(state + bytes([1 if okf else 0]) + bytes(7))
                    if rank == 0:
                        ok = ok and all(b[576] == 1 for b in blobs) and fv_cache.finalverify_shards([b[:576] for b in blobs])
        return ok

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    assert run_steps(max(a.warmup, 0), False), "warm-up batch must verify"
    sync()
    t0 = time.perf_counter()
    ok = run_steps(a.steps, True)
    sync()
    dt = time.perf_counter() - t0
    assert ok
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        all_ms = {k: v / a.steps for k, v in stage_acc.items()}
        stage_ms = {k: v for k, v in all_ms.items() if not k.startswith("k_")}
        kernel_ms = {k: v for k, v in all_ms.items() if k.startswith("k_") and k in KERNEL_BYTES}
        kernel_ms.update({KERNEL_OF_STAGE[k]: v for k, v in stage_ms.items() if k in KERNEL_OF_STAGE})
        dom = max(kernel_ms, key=lambda k: kernel_ms[k])                 # the dominant single kernel
        alg_bytes = KERNEL_BYTES[dom] * n
        achieved = alg_bytes / (kernel_ms[dom] * 1e-3) / 1e9 if kernel_ms[dom] > 0 else 0.0
        whole = 320.0 * n / (stage_ms["total"] * 1e-3) / 1e9
        out = {
            "metric": "BLS sig verifications/sec (batch)",
            "value": n_total * a.steps / dt,
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic: %d distinct valid (pk, SHA256('msg'+i), sig) tuples per GPU (made by the device signer), "
                    "rnd=SHA256('Mr F was here'), resident in HBM" % n,
            "config": {"workload": "BatchedBLSVerifier batchVerify, %d-tuple batch per GPU" % n, "global_batch": n_total,
                       "blinding_chains": nthreads, "parallelism": "shard%d" % world, "batches_in_flight": inflight},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "whole_path_GBs_at_320B_per_tuple": whole,
                         "note": "integer-ALU bound path (about 4e6 32x32+64-bit multiply-adds per tuple); see DESIGN.md section 4"},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "kernel_ms": {k: round(v, 3) for k, v in kernel_ms.items()},
            "stage_ms_note": "HIP-event durations per stage inside the timed region; with %d batches in flight they include "
                             "time shared with other batches' kernels" % inflight,
            "input_gen_s": round(gen_s, 1),
        }
        out["roofline"]["traffic"] = pmc_traffic(dom)
        if not a.no_aux and world == 1:
            out["aux"] = aux_rows(m, cache, dev)
        if not a.no_cpu and world == 1:                # the CPU baseline is timed at N = 1 only
            import c_oracle as co      # the CPU restatement: this leg only
            out["cpu_baseline"] = cpu_baseline(co, a.cpu_sample, bytes(rnd))
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_summary.json: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no profile is committed for that kernel."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        return d[kernel]["hbm_bytes_corrected"]
    except Exception:
        return None


R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def secret_key(i):
    """Deterministic synthetic secret scalar in [1, r)."""
    return int.from_bytes(hashlib.sha256(b"sk" + i.to_bytes(8, "little")).digest(), "little") % (R_ORDER - 1) + 1


def sign_records(m, cache, dev, ids, sks=None, msgs=None):
    """SignatureSet records resident in HBM: tuple i = (pk_i, SHA256("msg"+i), sig_i) (the generator of
    benchmarks/bls_signature.nim:258-268), produced by the device signer."""
    ids = list(ids)
    sks = sks if sks is not None else [secret_key(i) for i in ids]
    msgs = msgs if msgs is not None else [hashlib.sha256(b"msg" + str(i).encode()).digest() for i in ids]
    d_sk = torch.frombuffer(bytearray(b"".join(s.to_bytes(32, "little") for s in sks)), dtype=torch.uint8).to(dev)
    d_ms = torch.frombuffer(bytearray(b"".join(msgs)), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(320 * len(ids), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    ok, _ = m.signSets_device(cache, d_sk.data_ptr(), d_ms.data_ptr(), len(ids), d_out.data_ptr())
    assert ok
    return d_out


def aux_rows(m, cache, dev):
    """Side measurements of the other BASELINE.json configs (not the headline metric):
    config 3 fastAggregateVerify with 32 768 keys, config 4 G1 Pippenger MSM with 2^20 points."""
    import random
    import numpy as np
    out = {}
    n = 32768
    msg = hashlib.sha256(b"Mr F was here").digest()
    sks = [secret_key((1 << 41) + i) for i in range(n)]
    d_pks = sign_records(m, cache, dev, range(n), sks=sks, msgs=[msg] * n).view(n, 320)[:, :96].contiguous()
    # the aggregate signature of the n signers on msg is [sum sk_i mod r]H(msg): one more signer call
    sig = bytes(sign_records(m, cache, dev, [0], sks=[sum(sks) % R_ORDER], msgs=[msg]).cpu().numpy())[128:320]
    fav = lambda: m._check(m.lib().mi355_bls_fast_aggregate_verify_device(cache._h, d_pks.data_ptr(), n, msg, len(msg), sig, 0))
    assert fav() == 1
    t0 = time.perf_counter()
    for _ in range(3):
        assert fav() == 1
    t = cache.timings()
    out["fastAggregateVerify_32768"] = {"ms_per_call": (time.perf_counter() - t0) / 3 * 1e3, "g1_sum_ms": t["blinding"],
                                        "g1_sum_GBs_at_96B_per_key": 96.0 * n / (t["blinding"] * 1e-3) / 1e9,
                                        "note": "one pairing per call: latency-bound (wave-cooperative hash-to-G2, 2-pair Miller loop with 8 lanes per pair, final exponentiation)"}
    # config 2: 4 096-tuple batches (latency-bound: the kernels of one such batch fill a sixteenth of the chip)
    n4 = 4096
    d4 = sign_records(m, cache, dev, range(n4))
    rnd = hashlib.sha256(b"Mr F was here").digest()
    c4 = [m.BatchedBLSVerifierCache.init(max_sets=n4, device=dev.index or 0) for _ in range(8)]
    s4 = [torch.cuda.Stream(device=dev) for _ in range(8)]
    for c, st in zip(c4, s4):
        assert c.verify_device(d4.data_ptr(), n4, rnd, st.cuda_stream)
    t0 = time.perf_counter()
    for _ in range(5):
        assert c4[0].verify_device(d4.data_ptr(), n4, rnd, s4[0].cuda_stream)
    one = (time.perf_counter() - t0) / 5
    for c in c4:
        c.set_cooperative(False)                   # many batches in flight: one lane per set is the efficient form
    reps = 72
    for i in range(8):                             # warm-up of the other kernel variants
        c4[i].submit_device(d4.data_ptr(), n4, rnd, s4[i].cuda_stream)
    for i in range(8):
        assert c4[i].wait()
    t0 = time.perf_counter()
    for i in range(reps):
        if i >= 8:
            assert c4[i % 8].wait()
        c4[i % 8].submit_device(d4.data_ptr(), n4, rnd, s4[i % 8].cuda_stream)
    for i in range(8):
        assert c4[(reps + i) % 8].wait()
    dt8 = (time.perf_counter() - t0) / reps
    out["batchVerify_4096"] = {"ms_per_blocking_call": one * 1e3, "verifications_per_s_one_caller": n4 / one,
                               "verifications_per_s_8_in_flight": n4 / dt8}
    for c in c4:
        c.close()
    nm = 1 << 20
    rng = random.Random(7)
    base = sign_records(m, cache, dev, range(2048), sks=[rng.getrandbits(96) | 1 for _ in range(2048)], msgs=[msg] * 2048)
    dp = base.view(2048, 320)[:, :96].contiguous().repeat(nm // 2048, 1).reshape(-1)     # P_i = [a_i]G1, a_i 96-bit
    sc = np.random.default_rng(7).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    t0 = time.perf_counter()
    for _ in range(3):
        m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    dt = (time.perf_counter() - t0) / 3
    out["g1_msm_2^20"] = {"points_per_s": nm / dt, "ms_per_call": dt * 1e3, "nbits": 255,
                          "GBs_at_128B_per_point": 128.0 * nm / dt / 1e9}
    return out


def cpu_baseline(co, sample, rnd):
    """The C restatement of the reference algorithm (one pairing context per thread, parallel_chunks
    split, update loop, commit, merge, finalverify) on all host cores, bounded to roughly 10-20 s."""
    cores = co.lib().oracle_num_threads()
    probe = co.make_batch(2 * cores, seed=1 << 40)
    t0 = time.perf_counter()
    assert co.batch_verify(probe, rnd, cores)
    per = (time.perf_counter() - t0) / 2          # seconds for one tuple per core
    if sample <= 0:
        sample = max(2 * cores, min(65536, int(12.0 / per) * cores // 1))
    recs = co.make_batch(min(sample, 4096), seed=(1 << 40) + 7)
    if sample > 4096:
        recs = (recs * ((sample + 4095) // 4096))[:320 * sample]
    t0 = time.perf_counter()
    ok = co.batch_verify(recs, rnd, cores)
    dt = time.perf_counter() - t0
    assert ok
    # one core (batchVerifySerial shape), a few seconds
    n1 = min(1024, len(recs) // 320)
    t0 = time.perf_counter()
    assert co.batch_verify(recs[:320 * n1], rnd, 1)
    dt1 = time.perf_counter() - t0
    import ctypes.util
    return {"value": sample / dt, "unit": "verifications/s", "cores": cores, "kind": "port",
            "value_1core": n1 / dt1,
            "blst_on_this_box": bool(ctypes.util.find_library("blst")),      # use BLST itself if the box has it
            "sample": "%d tuples of the same workload, batchVerifyParallel shape with %d threads, %.1f s; 1 core: %d tuples, %.1f s "
                      "(oracle/bls_oracle.c: unoptimised plain-C restatement of the reference algorithm, not BLST; "
                      "BLST's assembly is several times faster per core)" % (sample, cores, dt, n1, dt1)}


if __name__ == "__main__":
    main()
