#!/usr/bin/env python3
"""Headline benchmark: BLS batch signature verifications / second on MI355X.

A step = one batchVerify of a 65 536-tuple batch per GPU (the size BASELINE.json's target is quoted on), inputs
resident in HBM before the timed region; `--inflight` (default 3) independent caller contexts keep that many
batches in flight so the serial tail of one overlaps the wide kernels of another.  N > 1: one process per GPU, each
verifies its own 65 536-tuple shard of one global batch (weak scaling); the only exchange is an RCCL all_gather of
the 640-byte shard blob (576-byte committed Fp12 state + ok word) per rank on DEVICE buffers, then one final exponentiation on
rank 0.  The blob itself never crosses PCIe; the host does wait for the shard (`shard_wait`: a host-side stream wait and a
580-byte D2H of the ok word) BEFORE it issues the collective, because torch enqueues collectives on a stream of its own and one
that waits there for a whole batch blocks the hardware queue it shares with a caller stream.

Prints ONE JSON line (rank 0) with the contract fields plus
  value_one_caller / value_host_buffers   one blocking caller, HBM-resident and PCIe-inclusive (outside the timed region)
  roofline      dominant kernel (chosen by kernel-alone durations of an un-overlapped profiling step), HBM bytes,
                and roofline.int_mad: the integer multiply-add roofline this path is actually bound by
  cpu_baseline  the C restatement of the reference algorithm (oracle/bls_oracle.c) on the host cores, with legs for
                the other BASELINE configs; BLST itself if the box has a libblst.
"""
import argparse
import ctypes
import ctypes.util
import hashlib
import json
import math
import os
import sys
import time

# HIP runtime: hardware queues per device (default 4).  Streams beyond that share a queue and their kernels serialise; a caller that
# keeps many SMALL batches in flight (config 2: 4 096-tuple batches, a few waves per kernel) is limited by that: 1.39 M/s with
# 4 queues, 2.3 M/s with 8.  Whole-chip batches (the headline) are not affected.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
# 32x32+64-bit multiply-adds (v_mad_i64_i32 / v_mad_u64_u32) per tuple and kernel of the one-lane-per-tuple pipeline:
# a census of the real formulas (tests/host_emu: emu_mad_census; tests/test_host_emu.py pins this table to it)
MAD_PER_TUPLE = {"k_hash_map": 680358, "k_hash_clear": 1079568, "k_pkmul": 259357, "k_sig_bucket": 87808, "k_lines": 735301,
                 "k_lineprod": 1119552}
MAD_ISSUE_CYCLES = 4.0          # one wave64 VALU instruction per SIMD per 4 cycles (MI355X_MICROARCH.md, issue cost table)
CLOCK_HZ = 2.4e9                # peak engine clock; under this load the chip sustains less (DVFS), see DESIGN.md section 4
MAD_PEAK_MEASURED = 256 * 4 * 64 / 2.28e-9      # multiply-adds/s the chip issues in the micro-benchmark (profiles/r01_ubench_valu.txt)
# The CEILING for this arithmetic: back-to-back lazily reduced dot products (588 multiply-adds + 69 other instructions each) with no
# caller code at all, every SIMD busy (tools/ubench_fp2chain.hip, profiles/r04_ubench_fp2chain.txt): 30.39 T multiply-adds/s - the
# chip holds 2.17 GHz under that stream and issues an instruction per 4.09 cycles
MAD_CEILING = 30.39e12          # the committed figure; the run measures it again on its own box (measure_ceiling)
# algorithmic HBM bytes per tuple of the WHOLE path, SURVEY.md section 8(d): one 320-byte SignatureSet read per verification
PATH_BYTES_PER_TUPLE = 320
# the fixed per-kernel table of the roofline object (the wide kernels of one batch, in pipeline order)
# one blocking batchVerify at these sizes, GPU and CPU (benchmarks/bench_all.nim:48-65 uses 6 / 60 / 180; 65 536 is the headline batch)
LATENCY_CURVE_SIZES = (1, 6, 60, 180, 512, 1024, 4096, 16384, 65536)
ROOFLINE_KERNELS = ("k_hash_map", "k_hash_clear", "k_pkmul", "k_sig_bucket", "k_lines", "k_lineprod")
# which stage timer (HIP events inside the library) measures which single kernel
KERNEL_OF_STAGE = {"pk_mul": "k_pkmul", "miller_lines": "k_lines"}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(a):
    """`bench.py --gpus N` without a launcher: this process starts N rank processes itself (one per GPU, torchrun-style
    environment) BEFORE it makes any GPU call of its own, relays rank 0's one JSON line and fails if any rank fails.
    (Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks already exist: WORLD_SIZE is set and this is skipped.)"""
    import subprocess
    n = a.gpus
    have = torch.cuda.device_count()                 # counts devices without initialising the GPU
    if have < n and os.environ.get("BENCH_ALL_ON_DEVICE0") != "1" and os.environ.get("BENCH_FORCE_LOCAL0") != "1":
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) are visible\n" % (n, have))
        return 2
    ge.build(load=False)                             # compile once here, not N times in the ranks
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    out0 = []
    th = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    th.start()
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in list(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        for r in live:                               # exactly the processes started above
            procs[r].kill()
        for pr in procs:
            pr.wait()
        sys.stderr.write("bench.py: rank %d exited with code %d\n" % failed)
        return 1
    th.join(10)
    lines = [l for l in "".join(out0).splitlines() if l.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def measure_ceiling():
    """tools/ubench_fp2chain.bin on THIS box (a child process, ~50 ms of GPU time before the timed region): back-to-back lazily reduced
    dot products with no caller code on every SIMD - the most any kernel built on this multiplier can reach.  None if the binary is
    missing or fails (the committed figure of profiles/r04_ubench_fp2chain.txt is then the fallback, and the line says so)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench_fp2chain.bin")
    try:
        out = subprocess.run([exe, "4000", "0"], capture_output=True, text=True, timeout=120).stdout
        mm = re.search(r"cycles/instr ([0-9.]+)\s+in-kernel clock ([0-9.]+) GHz\s+multiply-adds/s ([0-9.]+) T", out)
        return {"mads_per_s": float(mm.group(3)) * 1e12, "cycles_per_instr": float(mm.group(1)), "clock_ghz": float(mm.group(2))}
    except Exception:
        return None


def dist_info(world, backend, ctl, dev):
    """What stands behind an N > 1 line: backend, world size, RCCL version and the identity of every rank's device, gathered over the
    gloo control group - N distinct devices means N GPUs were really used."""
    try:
        prop = torch.cuda.get_device_properties(dev)
        ident = str(getattr(prop, "uuid", "")) or ""
    except Exception:
        ident = ""
    if not ident or set(ident) <= set("0-"):
        ident = "hip-device"
    ident = "%s/dev%d" % (ident, dev.index)
    objs = [None] * world
    if world > 1:
        dist.all_gather_object(objs, ident, group=ctl)
    else:
        objs = [ident]
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {"backend": backend, "is_rccl": backend == "nccl", "world_size": world, "nccl_version": ver, "device_uuids": objs,
            "distinct_devices": len(set(objs)), "all_on_device0_test_hook": os.environ.get("BENCH_ALL_ON_DEVICE0") == "1"}


def next_rnd(rnd):
    """secureRandomBytes of the next step: re-hashed before every iteration as benchmarks/bls_signature.nim:269-275 does."""
    return hashlib.sha256(rnd).digest()


class ShardedRun:
    """One configuration of the timed loop: `inflight` caller contexts on this rank's GPU, n tuples per GPU per step."""

    def __init__(self, m, a, dev, local, rank, world, backend, ctl, n, sharded_path):
        self.m, self.a, self.dev, self.rank, self.world, self.backend, self.ctl, self.n, self.sharded_path = m, a, dev, rank, world, backend, ctl, n, sharded_path
        # caller streams first: HIP spreads streams over its hardware queues as they are created, and two callers
        # whose streams share a hardware queue run strictly one after the other (seen with rocprofv3 --kernel-trace)
        self.inflight = inflight = max(1, a.inflight)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
        t0 = time.time()
        # n distinct valid (pk, SHA256("msg"+i), sig) tuples made by the library's own device signer
        # (mi355_bls_sign_sets_device; parity-tested against the oracle in tests/test_gpu_sign.py)
        gen = m.BatchedBLSVerifierCache.init(max_sets=n, device=local)
        self.d_sets = sign_records(m, gen, dev, range(rank * n, rank * n + n))
        gen.close()
        self.gen_s = time.time() - t0
        self.n_total = n * world
        self.nthreads = nthreads = m.DEFAULT_NUM_THREADS * world                # global number of blinding chains
        # `inflight` independent callers (one context + stream each, "one context per concurrent caller",
        # bls_batch_verifier.nim:389-391) keep several batches in flight so that one batch's serial tail
        # (step products, Horner, final exponentiation: a handful of waves) overlaps another batch's wide kernels.
        self.caches = [m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nthreads, device=local) for _ in range(inflight)]
        self.throughput_mode = a.ctx_mode == "throughput" or (a.ctx_mode == "auto" and inflight > 1)
        for c in self.caches:
            c.set_cooperative(not self.throughput_mode)  # throughput mode: several batches in flight (include/blscurve_mi355x.h)
        self.fv_caches = [m.BatchedBLSVerifierCache.init(max_sets=64, numThreads=nthreads, device=local) for _ in range(inflight)] if sharded_path else []
        self.lo, self.hi, first, count = m.shard_plan(self.n_total, nthreads, world, rank)
        assert count == n
        self.exchange = {"mode": a.exchange if backend == "nccl" else "host"}
        self.gathered = [torch.zeros(world * 640, dtype=torch.uint8, device=dev) for _ in range(inflight)] if sharded_path else []
        # every context writes its shard blob straight into a torch tensor: the send buffer of the collective
        self.mine_t = [torch.zeros(640, dtype=torch.uint8, device=dev) for _ in range(inflight)] if sharded_path else []
        for c, t in zip(self.caches, self.mine_t):
            c.set_shard_blob_ptr(t.data_ptr())
        self.stage_acc = {}
        self.fv_busy = [False] * inflight
        self.rnd_of_slot = [None] * inflight

    def close(self):
        for c in self.caches + self.fv_caches:
            c.close()
        self.d_sets = None

    def all_gather_host(self, blob):
        """host exchange: over the gloo control group, so that it still works when the RCCL transport is what failed"""
        mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        allst = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allst, mine, group=self.ctl)
        return [bytes(t.numpy().tobytes()) for t in allst]

    def record_timings(self, c):
        for kk, v in list(c.timings().items()) + list(c.kernel_timings().items()):
            self.stage_acc[kk] = self.stage_acc.get(kk, 0.0) + v

    def submit(self, slot, after, rnd):
        c, st = self.caches[slot], self.streams[slot]
        if not self.sharded_path:
            c.submit_device(self.d_sets.data_ptr(), self.n, rnd, st.cuda_stream, after=after)
        else:
            c.shard_submit_device(self.d_sets.data_ptr(), self.n_total, self.lo, self.hi, rnd, st.cuda_stream, after=after)

    def fv_drain(self, slot):
        """Verdict of the merge + final exponentiation enqueued for this slot one round ago (rank 0)."""
        if not self.fv_busy[slot]:
            return True
        self.fv_busy[slot] = False
        return self.fv_caches[slot].finalverify_wait()

    def collect(self, slot, record):
        c, st = self.caches[slot], self.streams[slot]
        if not self.sharded_path:
            res = c.wait()
        elif self.exchange["mode"] == "device":
            # The bench SYNCHRONISES on the shard (shard_wait: a host wait + 580-byte D2H) before it issues the collective: torch
            # enqueues collectives on a stream of its own, and a collective that had to wait there for a whole batch would block the
            # hardware queue it shares with a caller stream (measured: 18.3 instead of 13.5 ms per step when the all_gather is enqueued
            # right behind the submit).  The all_gather itself runs on device buffers (RCCL over xGMI); merge + final exponentiation
            # (rank 0) are enqueued behind it and their verdict is read one round later (fv_drain), so the host never waits for them
            # while other batches are in flight.
            res = self.fv_drain(slot)
            state, okf = c.shard_wait()
            with torch.cuda.stream(st):
                dist.all_gather_into_tensor(self.gathered[slot], self.mine_t[slot])
            if self.rank == 0:
                self.fv_caches[slot].finalverify_blobs_submit(self.gathered[slot].data_ptr(), self.world, 640, st.cuda_stream)
                self.fv_busy[slot] = True
        else:
            state, okf = c.shard_wait()
            blobs = self.all_gather_host(state + bytes([1 if okf else 0]) + bytes(7))
            res = True
            if self.rank == 0:
                res = all(b[576] == 1 for b in blobs) and self.fv_caches[slot].finalverify_shards([b[:576] for b in blobs])
        if record:
            self.record_timings(c)
        return res

    def run_steps(self, k, record, rnd):
        """One host thread keeps `inflight` batches in flight with the submit / wait entry points (context
        i % inflight; a context is waited for right before it is reused, i.e. oldest first).  Every batch is
        chained to the one submitted before it (`after`): it starts when that one has finished hashing and its public-key
        multiplications, so the batches in flight sit at different stages and a serial tail always runs beside whole-chip
        kernels of another batch.  secureRandomBytes is re-hashed before every step (benchmarks/bls_signature.nim:269-275).
        Returns (every verdict true, rnd after the last step)."""
        ok = True
        inflight, caches = self.inflight, self.caches
        busy = [False] * inflight
        for it in range(k):
            slot = it % inflight
            if busy[slot]:
                ok = self.collect(slot, record) and ok
            # chaining staggers whole-chip batches; small batches do not fill the chip and simply run side by side
            after = caches[(slot - 1) % inflight] if (inflight > 1 and self.n >= 32768) else None
            rnd = next_rnd(rnd)
            self.submit(slot, after, rnd)
            busy[slot] = True
        for j in range(inflight):
            slot = (k + j) % inflight
            if busy[slot]:
                ok = self.collect(slot, record) and ok
                busy[slot] = False
        for slot in range(inflight):
            ok = self.fv_drain(slot) and ok
        return ok, rnd

    def run_steps_threads(self, k, record, rnd):
        """--threads: one blocking call per host thread (N = 1 only)."""
        from concurrent.futures import ThreadPoolExecutor
        import queue
        free = queue.Queue()
        for i in range(self.inflight):
            free.put(i)
        rnds = []
        for _ in range(k):
            rnd = next_rnd(rnd)
            rnds.append(rnd)

        def one(r):
            slot = free.get()
            try:
                return self.caches[slot].verify_device(self.d_sets.data_ptr(), self.n, r, self.streams[slot].cuda_stream)
            finally:
                free.put(slot)
        with ThreadPoolExecutor(max_workers=self.inflight) as pool:
            return all(pool.map(one, rnds)), rnd

    def reset_after_transport_error(self):
        torch.cuda.synchronize()
        for i in range(self.inflight):
            self.fv_busy[i] = False
        for c in self.caches + self.fv_caches:
            for fn in (c.shard_wait, c.finalverify_wait):
                try:
                    fn()
                except Exception:
                    pass

    def timed(self, steps, warmup):
        """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; returns the max over ranks (s)."""
        a, world, backend, dev = self.a, self.world, self.backend, self.dev
        runner = self.run_steps_threads if (a.threads and not self.sharded_path) else self.run_steps
        rnd = hashlib.sha256(b"Mr F was here").digest()

        def sync():
            if self.sharded_path:
                dist.barrier()
            torch.cuda.synchronize()

        # Warm-up.  Only a TRANSPORT error of the device-side exchange (RCCL all_gather on device buffers) makes the run fall back
        # to the host exchange, and only after all ranks agreed on it over the gloo control group; a wrong verdict never does: it
        # fails the run.
        err = None
        ok = True
        try:
            ok, rnd_w = runner(max(warmup, 0), False, rnd)
        except (RuntimeError, dist.DistBackendError) as e:
            if not (self.sharded_path and self.exchange["mode"] == "device"):
                raise
            err = e
        if self.sharded_path and self.exchange["mode"] == "device":
            flag = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.ctl)
            if int(flag.item()) == 0:
                self.exchange["mode"] = "host"
                self.exchange["fallback"] = ("device-side exchange failed on this rank (%s: %s); host exchange used" % (type(err).__name__, str(err)[:200])
                                             if err is not None else "another rank's device-side exchange failed; host exchange used")
                self.reset_after_transport_error()
                ok, rnd_w = runner(max(warmup, 0), False, rnd)
        assert ok, "a warm-up batch did not verify"
        sync()
        t0 = time.perf_counter()
        ok, _ = runner(steps, True, rnd_w)
        sync()
        dt = time.perf_counter() - t0
        assert ok, "a timed batch did not verify"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if self.sharded_path:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())


def msm_rows_sharded(m, cache, dev, rank, world, backend, ctl, log2_points=20):
    """G1 MSM points/s on N GPUs (BASELINE.json's second metric), point-sharded: every rank computes the
    full-width partial of its points and leaves it in device memory, ONE all_gather of 144 bytes per rank, rank 0 adds the
    partials.  weak: 2^20 points per GPU (the config-4 shape per device); strong: 2^20 points in all."""
    import numpy as np
    out = {}
    nm = 1 << log2_points
    msg = hashlib.sha256(b"Mr F was here").digest()
    import random
    rng = random.Random(7 + rank)
    base = sign_records(m, cache, dev, range(2048), sks=[rng.getrandbits(96) | 1 for _ in range(2048)], msgs=[msg] * 2048)
    dp = base.view(2048, 320)[:, :96].contiguous().repeat(nm // 2048, 1).reshape(-1)     # P_i = [a_i]G1, a_i 96-bit
    ds = torch.frombuffer(bytearray(np.random.default_rng(7 + rank).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()), dtype=torch.uint8).to(dev)
    part = torch.zeros(144, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(144 * world, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)

    def one(npts):
        m.p1s_mult_pippenger_partial_device(cache, part.data_ptr(), dp.data_ptr(), npts, ds.data_ptr(), 255, st.cuda_stream)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, part)
        else:
            lst = [torch.empty(144, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(lst, part.cpu(), group=ctl)
            gathered.copy_(torch.cat(lst).to(dev))
        if rank == 0:
            return m.p1s_add_device(cache, gathered.data_ptr(), world, 144, st.cuda_stream)
        torch.cuda.synchronize()
        return None

    for name, npts in (("weak_2^%d_per_gpu" % log2_points, nm), ("strong_2^%d_total" % log2_points, m.msm_shard_range(nm, world, rank)[1])):
        res = one(npts)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            res = one(npts)
        dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tm = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=ctl)
        dt = float(tm.item())
        total = nm * world if name.startswith("weak") else nm
        out[name] = {"points_per_s": total / dt, "ms_per_call": dt * 1e3, "points_total": total, "nbits": 255, "result_is_point": res is None or len(res) == 144}
    return out


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
    ap.add_argument("--no-aux", action="store_true", help="skip the fastAggregateVerify / MSM / small-batch side measurements")
    ap.add_argument("--ctx-mode", choices=["auto", "latency", "throughput"], default="auto",
                    help="mode of the contexts of the timed region (mi355_bls_ctx_set_cooperative): auto = throughput when several batches are in flight")
    ap.add_argument("--no-one-caller", action="store_true", help="skip the one-blocking-caller measurements after the timed region (profiling runs)")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path (shards + collective) even with one rank")
    ap.add_argument("--msm-log2", type=int, default=20, help="N > 1: log2 of the points per GPU of the multi-GPU MSM rows (tests pass a small value)")
    ap.add_argument("--no-ceiling", action="store_true", help="do not run tools/ubench_fp2chain.bin before the timed region")
    ap.add_argument("--exchange", choices=["device", "host"], default=os.environ.get("BENCH_EXCHANGE", "device"),
                    help="N > 1: all_gather of device-resident shard blobs (RCCL, no host round trip) or of host bytes")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))                 # this process never touches the GPU: the ranks are its children

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N does so itself; under "
                         "torch.distributed.run pass --nproc-per-node N and --gpus N)\n" % (a.gpus, world))
        sys.exit(2)
    # BENCH_DIST_BACKEND=gloo + BENCH_ALL_ON_DEVICE0=1 is a TEST HOOK: it lets the N > 1 code path (sharding, exchange,
    # merge) run end to end on a box with a single GPU; real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_ALL_ON_DEVICE0") == "1" or os.environ.get("BENCH_FORCE_LOCAL0") == "1":
        local = 0                # BENCH_FORCE_LOCAL0: TEST HOOK of the guard below (ranks on one device WITHOUT declaring the dry run)
    sharded_path = world > 1 or a.force_dist
    ctl = None
    if sharded_path:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        # control group over gloo: agreement on the exchange mode and the host exchange must not depend on the RCCL transport
        ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    ge.build()
    m = ge.load_package()
    binfo = m.build_info()
    if not binfo["aligned"] and os.environ.get("BLS_NO_ALIGN") != "1":
        sys.stderr.write("bench.py: the library was built WITHOUT the instruction-alignment post-pass (%s); numbers from it are not the product's. "
                         "Rebuild (nim-blscurve_amd/build.sh -f) or set BLS_NO_ALIGN=1 to measure it on purpose.\n" % binfo)
        sys.exit(2)
    ceiling_live = measure_ceiling() if (rank == 0 and not a.no_ceiling) else None
    dinfo = dist_info(world, backend, ctl, dev) if sharded_path else None
    if dinfo and world > 1 and not dinfo["all_on_device0_test_hook"] and dinfo["distinct_devices"] != world:
        # a real N-GPU run whose ranks landed on fewer than N devices would report a scaling curve of one GPU: fail loudly instead
        sys.stderr.write("bench.py: %d ranks but %d distinct devices (%s); set BENCH_ALL_ON_DEVICE0=1 only for the documented dry run\n"
                         % (world, dinfo["distinct_devices"], dinfo["device_uuids"]))
        sys.exit(2)

    n = a.batch
    run = ShardedRun(m, a, dev, local, rank, world, backend, ctl, n, sharded_path)
    inflight, streams, caches, d_sets = run.inflight, run.streams, run.caches, run.d_sets
    n_total, nthreads, lo, hi, throughput_mode = run.n_total, run.nthreads, run.lo, run.hi, run.throughput_mode
    rnd = hashlib.sha256(b"Mr F was here").digest()
    cache = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nthreads, device=local)      # the one blocking caller: latency mode
    if os.environ.get("BENCH_TEST_DIE_RANK") == str(rank) and world > 1:
        # TEST HOOK (tests/test_gpu_bench_multirank.py): this rank dies after its set-up, the others are left inside the first
        # exchange - spawn_ranks must notice, end them and exit non-zero
        os._exit(3)
    dt = run.timed(a.steps, a.warmup)
    exchange, stage_acc, gen_s = run.exchange, run.stage_acc, run.gen_s

    out = None
    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        all_ms = {k: v / a.steps for k, v in stage_acc.items()}
        stage_ms = {k: v for k, v in all_ms.items() if not k.startswith("k_")}
        timed_kernel_ms = {k: v for k, v in all_ms.items() if k.startswith("k_") and k in KERNEL_BYTES}
        timed_kernel_ms.update({KERNEL_OF_STAGE[k]: v for k, v in stage_ms.items() if k in KERNEL_OF_STAGE})
        # ---- outside the timed region: ONE caller, un-overlapped (kernel-alone durations, single-caller rate, PCIe-inclusive rate)
        if a.no_one_caller:
            one = {"kernel_alone_ms": {**timed_kernel_ms, "k_sig_bucket": stage_ms.get("sig_mul_sum", 0.0)}, "tail_ms_alone": {}, "value_one_caller": None,
                   "ms_one_caller": None}
        else:
            if throughput_mode:
                cache_tp = caches[0]
            else:
                cache_tp = m.BatchedBLSVerifierCache.init(max_sets=n, numThreads=nthreads, device=local)
                cache_tp.set_cooperative(False)
            one = one_caller_rows(m, cache, cache_tp, streams[0], d_sets, n, n_total, lo, hi, rnd, sharded_path)
        alone = one["kernel_alone_ms"]
        if not any(k in KERNEL_BYTES for k in alone):              # --threads with --no-one-caller records no stage timers
            alone = {"k_lineprod": ms_per_step}
        prop = torch.cuda.get_device_properties(dev)
        mad_total = sum(MAD_PER_TUPLE.values())
        mad_peak = prop.multi_processor_count * 4 * 64 * CLOCK_HZ / MAD_ISSUE_CYCLES
        mad_achieved = mad_total * n / (ms_per_step * 1e-3)
        # One definition, the same every round: the WHOLE path at SURVEY.md 8(d)'s 320 bytes per tuple over the step time against the
        # HBM peak (a unit = one verification, a "launch" = one batch step), and a FIXED per-kernel table beside it - no "dominant
        # kernel" whose choice changes the number.  The bound that matters is the integer multiply-add issue rate (int_mad below).
        achieved = PATH_BYTES_PER_TUPLE * n / (ms_per_step * 1e-3) / 1e9
        ktab, traffic_total, traffic_src = {}, 0.0, None
        for k in ROOFLINE_KERNELS:
            tr, src = pmc_traffic(k)
            traffic_src = traffic_src or src
            ms_alone = alone.get(k)
            row = {"algorithmic_bytes": KERNEL_BYTES.get(k, 0) * n, "mad_per_tuple": MAD_PER_TUPLE.get(k), "hbm_traffic_bytes": tr,
                   "ms_alone": round(ms_alone, 4) if ms_alone else None, "ms_timed_region": round(timed_kernel_ms[k], 4) if k in timed_kernel_ms else None}
            if ms_alone:
                row["algorithmic_GBs_alone"] = KERNEL_BYTES.get(k, 0) * n / (ms_alone * 1e-3) / 1e9
                row["hbm_util_alone"] = (tr / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None
                row["int_mad_frac_alone"] = MAD_PER_TUPLE[k] * n / (ms_alone * 1e-3) / mad_peak
            if tr:
                traffic_total += tr
            ktab[k] = row
        ceil_v = ceiling_live["mads_per_s"] if ceiling_live else MAD_CEILING
        out = {
            "metric": "BLS sig verifications/sec (batch)",
            "value": n_total * a.steps / dt,
            "unit": "verifications/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic: %d distinct valid (pk, SHA256('msg'+i), sig) tuples per GPU (made by the device signer), "
                    "rnd=SHA256('Mr F was here') re-hashed before every step, resident in HBM" % n,
            "config": {"workload": "BatchedBLSVerifier batchVerify, %d-tuple batch per GPU" % n, "global_batch": n_total,
                       "blinding_chains": nthreads, "parallelism": "shard%d" % world, "batches_in_flight": inflight,
                       "context_mode": "throughput" if throughput_mode else "latency",
                       "exchange": exchange if sharded_path else None},
            "value_one_caller": one["value_one_caller"],
            "ms_one_caller": one["ms_one_caller"],
            "stage_ms_one_caller": one.get("stage_ms_one_caller"),
            "value_host_buffers": one.get("value_host_buffers"),
            "ms_host_buffers": one.get("ms_host_buffers"),
            "value_one_caller_sliced": one.get("value_one_caller_sliced"),
            "ms_one_caller_sliced_2^20": one.get("ms_one_caller_sliced_2^20"),
            "roofline": {"bound": "int_mad", "bound_note": "of the contract's two choices (hbm | mfma) this is the hbm line: achieved / peak / frac below are HBM "
                                                           "figures; what actually bounds the path is the integer multiply-add issue rate, int_mad",
                         "kernel": "whole path (one batch step: k_blind .. k_tail)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_unit": PATH_BYTES_PER_TUPLE, "units_per_launch": n,
                         "traffic": traffic_total or None, "traffic_source": traffic_src,
                         "traffic_note": "sum over the six wide kernels of their PMC-measured HBM bytes per launch (FETCH_SIZE doubled per the gfx950 note, + WRITE_SIZE)",
                         "kernels": ktab,
                         "int_mad": {"achieved": mad_achieved / 1e12, "peak": mad_peak / 1e12, "unit": "T multiply-adds/s",
                                     "frac": mad_achieved / mad_peak, "mad_per_tuple": mad_total,
                                     "model": "census of the kernels' formulas (MAD_PER_TUPLE) x tuples / ms_per_step; peak = CUs x 4 SIMDs x 64 lanes x "
                                              "2.4 GHz / 4 cycles per v_mad_i64_i32",
                                     "ceiling": {"achievable": ceil_v / 1e12, "frac_of_peak": ceil_v / mad_peak, "achieved_over_ceiling": mad_achieved / ceil_v,
                                                 "measured_on_this_box": ceiling_live is not None,
                                                 "cycles_per_instr": ceiling_live["cycles_per_instr"] if ceiling_live else None,
                                                 "clock_ghz": ceiling_live["clock_ghz"] if ceiling_live else None,
                                                 "source": "tools/ubench_fp2chain.bin run by this process before the timed region: back-to-back dot-product bodies, zero "
                                                           "caller code, one wave per SIMD on every SIMD" if ceiling_live else
                                                           "profiles/r04_ubench_fp2chain.txt (the binary did not run here)"},
                                     "peak_measured": MAD_PEAK_MEASURED / 1e12,
                                     "frac_of_measured": mad_achieved / MAD_PEAK_MEASURED,
                                     "peak_measured_note": "tools/ubench_valu.hip on this chip (profiles/r01_ubench_valu.txt): a stream of independent "
                                                           "v_mad_u64_u32 issues one per 2.28 ns per SIMD at 4 waves per SIMD (2.41 at 8, 2.87 at 1), "
                                                           "not one per 4 cycles at 2.4 GHz = 1.67 ns"},
                         "note": "the path is integer multiply-add bound, not HBM bound (1.2e4 multiply-adds per input byte): int_mad is the roofline "
                                 "that says how good the kernels are; the HBM fraction (320 B per verification over the step time) is what the contract asks for"},
            "build": binfo,
            "dist": dinfo,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "kernel_ms_timed_region": {k: round(v, 3) for k, v in timed_kernel_ms.items()},
            "kernel_ms_alone": {k: round(v, 3) for k, v in alone.items()},
            "tail_ms_alone": {k: round(v, 3) for k, v in one["tail_ms_alone"].items()},
            "stage_ms_note": "stage_ms / kernel_ms_timed_region: HIP-event durations inside the timed region (with %d batches in flight they include "
                             "time shared with other batches' kernels); kernel_ms_alone / tail_ms_alone: ONE blocking caller of a throughput-mode context after the "
                             "timed region (every kernel alone on the chip); ms_one_caller / stage_ms_one_caller: one blocking caller of a latency-mode "
                             "context (fork streams: its stages overlap)" % inflight,
            "input_gen_s": round(gen_s, 1),
        }
        if not a.no_aux and world == 1 and not a.force_dist:
            out["aux"] = aux_rows(m, cache, dev)
    # ---- N > 1, outside the timed region: the other multi-GPU rows (every rank takes part)
    if world > 1 and not a.no_aux:
        run.close()
        del run, caches, d_sets
        torch.cuda.synchronize()
        multi = {"g1_msm": msm_rows_sharded(m, cache, dev, rank, world, backend, ctl, a.msm_log2)}
        # STRONG scaling beside the weak headline (BASELINE's metric is quoted "at 1/2/4/8 MI355X"): ONE --batch-sized batch in all, batch / N
        # tuples per GPU, same exchange and merge.  (Small shards are latency-bound: a GPU verifies 8 192 tuples in ~5.7 ms, 65 536 in ~13.)
        ns = max(1, a.batch // world)
        runs = ShardedRun(m, a, dev, local, rank, world, backend, ctl, ns, True)
        ks = max(2, min(a.steps, 20)) if a.batch < 65536 else max(6, min(a.steps, 20))
        dts = runs.timed(ks, min(3, a.warmup))
        multi["batchVerify_strong"] = {"verifications_per_s": runs.n_total * ks / dts, "ms_per_step": dts / ks * 1e3, "global_batch": runs.n_total,
                                       "tuples_per_gpu": ns, "steps": ks, "scaling": "strong", "exchange": runs.exchange}
        runs.close()
        del runs
        if world == 8:
            # BASELINE config 5: 2^20 tuples across 8 GPUs = 131 072 per GPU = twice the headline batch (the row follows --batch so that a
            # dry run of this code path with small batches stays cheap: tests/test_gpu_bench_multirank.py)
            n5 = 2 * a.batch
            run5 = ShardedRun(m, a, dev, local, rank, world, backend, ctl, n5, True)
            k5 = max(2, min(a.steps, 20)) if a.batch < 65536 else max(6, min(a.steps, 20))
            dt5 = run5.timed(k5, min(3, a.warmup))
            multi["config5_batchVerify_2^20" if n5 == 131072 else "config5_shape_batchVerify"] = {
                "verifications_per_s": run5.n_total * k5 / dt5, "ms_per_step": dt5 / k5 * 1e3, "global_batch": run5.n_total, "tuples_per_gpu": n5, "steps": k5,
                "exchange": run5.exchange}
            run5.close()
        if rank == 0:
            out["multi_gpu"] = multi
    if rank == 0:
        if not a.no_cpu and world == 1 and not a.force_dist:                # the CPU baseline is timed at N = 1 only
            import c_oracle as co      # the CPU restatement: this leg only
            out["cpu_baseline"] = cpu_baseline(co, a.cpu_sample, rnd)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
            if "aux" in out and "latency_curve" in out["aux"]:
                out["crossover"] = crossover(out["aux"]["latency_curve"], out["cpu_baseline"])
        flatten_for_driver(out)
        print(json.dumps(out), flush=True)
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


HOT_KERNELS = ("k_hash_map", "k_hash_clear", "k_pkmul", "k_lines", "k_lineprod", "k_sig_bucket")
LATENCY_KERNELS = ("k_hash_one", "k_hash_map_rows", "k_team_clear", "k_team_clear_spread", "k_team_clear_rows", "k_team_lines", "k_team_lines_spread", "k_team_lines_rows",
                   "k_pip_rowtail", "k_tail", "k_fold")


def kernel_meta():
    """register / spill table build.sh leaves beside the library (tools/kernel_metadata.py --json), or None"""
    p = os.path.join(ROOT, "nim-blscurve_amd", "libblscurve_mi355x.so.kmeta.json")
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return None


def flatten_for_driver(out):
    """The driver's record keeps `config` and the FLAT scalars of `roofline` / `cpu_baseline` and drops every other key and every nested object
    (BENCH_r05.json: extra_keys).  The figures the review reads - the bound that matters, its ceiling, BASELINE's second metric, the latency
    rows - are therefore repeated here as flat scalars in the three objects that survive.  Nothing new is measured."""
    rf, cfg = out["roofline"], out["config"]
    im = rf.get("int_mad") or {}
    ce = im.get("ceiling") or {}
    rf["int_mad_frac"] = im.get("frac")
    rf["int_mad_tmads"] = im.get("achieved")
    rf["int_mad_peak_tmads"] = im.get("peak")
    rf["int_mad_over_ceiling"] = ce.get("achieved_over_ceiling")
    rf["ceiling_tmads"] = ce.get("achievable")
    rf["ceiling_clock_ghz"] = ce.get("clock_ghz")
    rf["ceiling_measured_on_this_box"] = ce.get("measured_on_this_box")
    km = kernel_meta()
    if km:
        # spills inside the assembly statements are zero by construction; what the metadata counts for the four assembly kernels is their
        # compiled prologue / flagged-lane fallback, so the figure is an upper bound of what the hot loops touch
        rf["vgpr_spill_max_hot"] = max((km[k]["spill"] or 0) for k in HOT_KERNELS if k in km)
        rf["vgpr_spill_max_latency"] = max([(v["spill"] or 0) for k, v in km.items() if k.split("<")[0] in LATENCY_KERNELS] or [0])
    for k, row in (rf.get("kernels") or {}).items():
        if row.get("ms_alone"):
            rf["ms_alone_" + k] = row["ms_alone"]
    cfg["ms_one_caller"] = out.get("ms_one_caller")
    cfg["ms_one_caller_sliced_2pow20"] = out.get("ms_one_caller_sliced_2^20")
    cfg["build_stamp"] = (out.get("build") or {}).get("stamp")
    cfg["build_aligned"] = (out.get("build") or {}).get("aligned")
    aux = out.get("aux") or {}
    msm = aux.get("g1_msm_2^20") or {}
    cfg["msm_points_per_s"] = msm.get("points_per_s")
    cfg["msm_ms_per_call"] = msm.get("ms_per_call")
    cfg["msm_ms_two_in_flight"] = msm.get("ms_per_msm_two_in_flight")
    cfg["fav_32768_ms"] = (aux.get("fastAggregateVerify_32768") or {}).get("ms_per_call")
    cfg["verify_one_signature_ms"] = (aux.get("verify_one_signature") or {}).get("ms_per_call")
    cfg["batch_64_ms"] = (aux.get("batchVerify_64") or {}).get("ms_per_blocking_call")
    cfg["batch_4096_ms"] = (aux.get("batchVerify_4096") or {}).get("ms_per_blocking_call")
    cfg["batch_4096_vps_16_in_flight"] = (aux.get("batchVerify_4096") or {}).get("verifications_per_s_16_in_flight")
    cfg["from_bytes_65536_ms"] = (aux.get("fromBytes_65536") or {}).get("ms_per_call")
    lc = aux.get("latency_curve") or []
    if lc:
        cfg["latency_floor_ms"] = min(r["ms_per_blocking_call"] for r in lc)
    cb = out.get("cpu_baseline")
    if cb:
        co = out.get("crossover") or {}
        cb["speedup_vs_cpu_port"] = out.get("speedup_vs_cpu_port")
        cb["gpu_faster_than_blst_model_from_n"] = co.get("gpu_faster_than_blst_model_from_n")
        # the stated MODEL of BLST (not a measurement: BLST is not on the box): 400 us per set per core + 600 us per call, at this box's cores
        cb["blst_model_vps"] = cb.get("cores", 1) / 400e-6 if not cb.get("blst_on_this_box") else None
        cb["value_over_blst_model"] = (out["value"] / cb["blst_model_vps"]) if cb.get("blst_model_vps") else None
    mg = out.get("multi_gpu") or {}
    for key, row in mg.items():
        if isinstance(row, dict):
            for k2, v2 in row.items():
                if isinstance(v2, (int, float)):
                    cfg["multi_%s_%s" % (key, k2)] = v2
                elif isinstance(v2, dict):
                    for k3, v3 in v2.items():
                        if isinstance(v3, (int, float)):
                            cfg["multi_%s_%s_%s" % (key, k2, k3)] = v3


def one_caller_rows(m, cache, cache_tp, stream, d_sets, n, n_total, lo, hi, rnd, sharded_path):
    """ONE blocking caller after the timed region: the single-caller rate ('kernels-only' with inputs resident) on
    the latency-mode context `cache`; kernel-alone durations from ONE blocking caller of a throughput-mode context `cache_tp`
    (one lane per item, no fork stream: every kernel has the chip to itself between its two events); the PCIe-inclusive rate
    of the host-buffer entry point."""
    out = {}
    reps = 5

    def caller(c):
        return (lambda: c.verify_device(d_sets.data_ptr(), n, rnd, stream.cuda_stream)) if not sharded_path else \
               (lambda: c.shard_device(d_sets.data_ptr(), n_total, lo, hi, rnd, stream.cuda_stream)[1])
    call = caller(cache)
    assert call()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert call()
    dt = (time.perf_counter() - t0) / reps
    out["ms_one_caller"] = dt * 1e3
    out["value_one_caller"] = n / dt
    out["stage_ms_one_caller"] = {k: round(v, 3) for k, v in cache.timings().items()}
    acc = {}
    call_tp = caller(cache_tp)
    assert call_tp()
    for _ in range(reps):
        assert call_tp()
        for k, v in list(cache_tp.timings().items()) + list(cache_tp.kernel_timings().items()):
            acc[k] = acc.get(k, 0.0) + v / reps
    alone = {k: v for k, v in acc.items() if k in KERNEL_BYTES}
    alone.update({KERNEL_OF_STAGE[k]: v for k, v in acc.items() if k in KERNEL_OF_STAGE})
    alone["k_sig_bucket"] = acc.get("sig_mul_sum", 0.0)
    out["kernel_alone_ms"] = {k: v for k, v in alone.items() if v > 0}
    out["tail_ms_alone"] = {"fold_of_line_products": acc.get("k_lineprod2", 0.0), "final": acc.get("final", 0.0)}      # a lone caller folds with k_fold
    if not sharded_path:
        host = d_sets.cpu().numpy().tobytes()                      # pageable host memory, as a Nim seq would be
        assert m.batchVerifyParallel(cache, host, rnd)
        t0 = time.perf_counter()
        for _ in range(reps):
            assert m.lib().mi355_bls_batch_verify(cache._h, host, n, rnd) == 1
        dt = (time.perf_counter() - t0) / reps
        out["ms_host_buffers"] = dt * 1e3
        out["value_host_buffers"] = n / dt
        # ONE blocking call on a batch 16 x the context's capacity (the reference's cache accepts any input.len,
        # bls_batch_verifier.nim:108-119,141): the library runs the slices pipelined over three internal workspaces, so the
        # blocking caller gets the pipelined rate, not the one-caller rate.  The 16 copies of the resident batch form one valid
        # batch of 16 n tuples (every tuple keeps its own blinding scalar).
        if n == 65536:
            try:
                big = d_sets.repeat(16)
                assert cache_tp.verify_device(big.data_ptr(), 16 * n, rnd, stream.cuda_stream)         # warm-up: creates the lanes
                t0 = time.perf_counter()
                for _ in range(2):
                    assert cache_tp.verify_device(big.data_ptr(), 16 * n, rnd, stream.cuda_stream)
                dt = (time.perf_counter() - t0) / 2
                out["ms_one_caller_sliced_2^20"] = dt * 1e3
                out["value_one_caller_sliced"] = 16 * n / dt
                del big
            except torch.OutOfMemoryError:
                out["value_one_caller_sliced"] = None
    return out


def pmc_traffic(kernel):
    """(HBM bytes per launch of the dominant kernel, file it was read from): the NEWEST committed rocprofv3 --pmc summary
    (profiles/rNN_pmc_summary.json, highest NN: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  (None, None) when no profile is committed for that kernel."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")) if re.search(r"r(\d+)_pmc_summary\.json$", f)]
    for f in sorted(files, key=lambda f: int(re.search(r"r(\d+)_pmc_summary\.json$", f).group(1)), reverse=True):
        try:
            d = json.load(open(f))
            return d[kernel]["hbm_bytes_corrected"], "profiles/" + os.path.basename(f)
        except Exception:
            continue
    return None, None


R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def g2_jacobian_image_to_affine(p288):
    """blst_p2 image (Jacobian X, Y, Z over Fp2, Montgomery R = 2^384) -> blst_p2_affine image (192 B).  Plain integers: input
    preparation of the aggregateVerify row."""
    rinv = pow(1 << 384, -1, P_MOD)
    f = [int.from_bytes(p288[48 * i:48 * i + 48], "little") * rinv % P_MOD for i in range(6)]
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P_MOD, (a[0] * b[1] + a[1] * b[0]) % P_MOD)
    X, Y, Z = (f[0], f[1]), (f[2], f[3]), (f[4], f[5])
    nz = pow((Z[0] * Z[0] + Z[1] * Z[1]) % P_MOD, -1, P_MOD)
    zi = (Z[0] * nz % P_MOD, -Z[1] * nz % P_MOD)
    zi2 = mul(zi, zi)
    x, y = mul(X, zi2), mul(Y, mul(zi2, zi))
    return b"".join((v * (1 << 384) % P_MOD).to_bytes(48, "little") for v in (x[0], x[1], y[0], y[1]))


def g1_jacobian_image_to_affine(p144):
    """blst_p1 image (Jacobian, Montgomery R = 2^384) -> affine (x, y) as plain integers; None for the point at infinity."""
    rinv = pow(1 << 384, -1, P_MOD)
    X, Y, Z = (int.from_bytes(p144[48 * i:48 * i + 48], "little") * rinv % P_MOD for i in range(3))
    if Z == 0:
        return None
    zi = pow(Z, -1, P_MOD)
    return (X * zi * zi % P_MOD, Y * zi * zi * zi % P_MOD)


def compress_records(recs):
    """SignatureSet records (blst Montgomery images, R = 2^384) -> ZCash compressed keys (48 B) and signatures (96 B):
    big-endian x with the flag bits 0x80 (compressed) and 0x20 (y is the lexicographically larger root).  Host-side input
    preparation of the fromBytes row (plain integers, no library code)."""
    rinv = pow(1 << 384, -1, P_MOD)
    fe = lambda b: int.from_bytes(b, "little") * rinv % P_MOD
    pks, sgs = [], []
    for i in range(len(recs) // 320):
        r = recs[320 * i:320 * i + 320]
        x, y = fe(r[0:48]), fe(r[48:96])
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if y > (P_MOD - 1) // 2 else 0)
        pks.append(bytes(b))
        x0, x1, y0, y1 = fe(r[128:176]), fe(r[176:224]), fe(r[224:272]), fe(r[272:320])
        big = (y1 > (P_MOD - 1) // 2) if y1 else (y0 > (P_MOD - 1) // 2)
        b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if big else 0)
        sgs.append(bytes(b))
    return b"".join(pks), b"".join(sgs)


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
    config 2 4 096-tuple batches, config 3 fastAggregateVerify with 32 768 keys, config 4 G1 Pippenger MSM with 2^20 points,
    plus the latency of the smallest calls (one signature; a 64-set batch)."""
    import random
    import numpy as np
    out = {}
    n = 32768
    msg = hashlib.sha256(b"Mr F was here").digest()
    sks = [secret_key((1 << 41) + i) for i in range(n)]
    d_pks = sign_records(m, cache, dev, range(n), sks=sks, msgs=[msg] * n).view(n, 320)[:, :96].contiguous()
    # the aggregate signature of the n signers on msg is [sum sk_i mod r]H(msg): one more signer call
    sig = bytes(sign_records(m, cache, dev, [0], sks=[sum(sks) % R_ORDER], msgs=[msg]).cpu().numpy())[128:320]
    fav = lambda k=n: m._check(m.lib().mi355_bls_fast_aggregate_verify_device(cache._h, d_pks.data_ptr(), k, msg, len(msg), sig, 0))
    assert fav() == 1
    t0 = time.perf_counter()
    for _ in range(5):
        assert fav() == 1
    t = cache.timings()
    out["fastAggregateVerify_32768"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3, "g1_sum_ms": t["blinding"], "stage_ms": {k: round(v, 4) for k, v in t.items()},
                                        "g1_sum_GBs_at_96B_per_key": 96.0 * n / (t["blinding"] * 1e-3) / 1e9,
                                        "note": "one pairing per call: latency-bound (wave-cooperative hash-to-G2, 2-pair Miller loop with 8 lanes per pair, final exponentiation)"}
    # config 1 shape on the device: ONE (pk, msg, sig) verification = fastAggregateVerify with one key (latency of the whole pipeline)
    one = sign_records(m, cache, dev, [0], sks=[sks[0]], msgs=[msg])
    pk1, sig1 = one[:96].contiguous(), bytes(one.cpu().numpy())[128:320]
    v1 = lambda: m._check(m.lib().mi355_bls_fast_aggregate_verify_device(cache._h, pk1.data_ptr(), 1, msg, len(msg), sig1, 0))
    assert v1() == 1
    t0 = time.perf_counter()
    for _ in range(5):
        assert v1() == 1
    out["verify_one_signature"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3}
    rnd = hashlib.sha256(b"Mr F was here").digest()
    # row f1: PublicKey.fromBytes + Signature.fromBytes for every tuple of a 65 536-tuple batch (compressed wire format resident in
    # HBM): decompression (one Fp / one Fp2 square root), "not infinity", both subgroup checks
    nd = 65536
    recs = bytes(sign_records(m, cache, dev, range(4096)).cpu().numpy())
    pk48, sg96 = compress_records(recs)
    d_pk = torch.frombuffer(bytearray(pk48 * (nd // 4096)), dtype=torch.uint8).to(dev)
    d_sg = torch.frombuffer(bytearray(sg96 * (nd // 4096)), dtype=torch.uint8).to(dev)
    d_ms = torch.frombuffer(bytearray(b"".join(recs[320 * i + 96:320 * i + 128] for i in range(4096)) * (nd // 4096)), dtype=torch.uint8).to(dev)
    back = ctypes.create_string_buffer(320 * 4096)
    deser = lambda out: m._check(m.lib().mi355_bls_deserialize_sets_device(cache._h, d_pk.data_ptr(), d_ms.data_ptr(), d_sg.data_ptr(), nd, 0, out, None))
    assert m._check(m.lib().mi355_bls_deserialize_sets_device(cache._h, d_pk.data_ptr(), d_ms.data_ptr(), d_sg.data_ptr(), 4096, 0, back, None)) == 1
    assert back.raw == recs                                            # the records the signer produced, byte for byte
    assert deser(None) == 1
    acc = 0.0
    for _ in range(5):
        assert deser(None) == 1
        acc += cache.timings()["total"] / 5
    out["fromBytes_65536"] = {"ms_per_call": acc, "tuples_per_s": nd / (acc * 1e-3),
                              "note": "kernel time (k_deser), compressed keys / messages / signatures and the 320-byte records resident in HBM"}
    # row f2: MultiSignatureSet.combine of 4 096 signatures on one message = a G1 and a G2 Pippenger run with 64-bit scalars (host arrays in)
    nc = 4096
    same = bytes(sign_records(m, cache, dev, range(nc), msgs=[msg] * nc).cpu().numpy())
    cpk = b"".join(same[320 * i:320 * i + 96] for i in range(nc))
    csg = b"".join(same[320 * i + 128:320 * i + 320] for i in range(nc))
    opk, osg = ctypes.create_string_buffer(96), ctypes.create_string_buffer(192)
    comb = lambda: m._check(m.lib().mi355_bls_combine(cache._h, rnd, cpk, csg, nc, opk, osg))
    comb()
    t0 = time.perf_counter()
    for _ in range(5):
        comb()
    out["combine_4096"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3}
    assert m.batchVerify(cache, opk.raw + msg + osg.raw, rnd) is True          # the combined set verifies
    # row f4: aggregateVerify of ONE aggregate signature over 1 024 (public key, message) pairs with distinct messages
    na = 1024
    ra = bytes(sign_records(m, cache, dev, range(na)).cpu().numpy())
    apk = [ra[320 * i:320 * i + 96] for i in range(na)]
    ams = [ra[320 * i + 96:320 * i + 128] for i in range(na)]
    asum = m.blst_p2s_mult_pippenger(b"".join(ra[320 * i + 128:320 * i + 320] for i in range(na)), b"\x01" * na, 8)      # sum of the signatures
    asig = g2_jacobian_image_to_affine(asum)
    assert m.aggregateVerify(cache, apk, ams, asig) is True
    t0 = time.perf_counter()
    for _ in range(5):
        assert m.aggregateVerify(cache, apk, ams, asig) is True
    out["aggregateVerify_1024"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3, "note": "host arrays in (Python list handling included)"}
    # row f3: the batch signer (publicFromSecret + coreSign per tuple; variable time: inputs of tests and benches only)
    sk_t = torch.frombuffer(bytearray(b"".join(secret_key(i).to_bytes(32, "little") for i in range(65536))), dtype=torch.uint8).to(dev)
    ms_t = torch.frombuffer(bytearray(b"".join(hashlib.sha256(b"msg" + str(i).encode()).digest() for i in range(65536))), dtype=torch.uint8).to(dev)
    o_t = torch.zeros(320 * 65536, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    oks, _ = m.signSets_device(cache, sk_t.data_ptr(), ms_t.data_ptr(), 65536, o_t.data_ptr())
    out["batch_signer_65536"] = {"ms_per_call": (time.perf_counter() - t0) * 1e3}
    assert oks
    # Latency curve: ONE blocking batchVerify per size, the sizes of the reference's own benchmark (benchmarks/bench_all.nim:48-65: batches of
    # 6 / 60 / 180 signatures) up to the headline batch, each on a context of its own size (latency mode, the default stream).  The CPU leg
    # (cpu_baseline.latency_curve) times the same sizes on all host cores; main() joins the two and names the crossover.
    dcurve = sign_records(m, cache, dev, range(1 << 23, (1 << 23) + 65536))
    curve = []
    for ncv in LATENCY_CURVE_SIZES:
        cc = m.BatchedBLSVerifierCache.init(max_sets=max(ncv, 1), device=dev.index or 0)
        for _ in range(2):
            assert cc.verify_device(dcurve.data_ptr(), ncv, rnd, 0)
        reps = 8 if ncv <= 4096 else 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            assert cc.verify_device(dcurve.data_ptr(), ncv, rnd, 0)
        dtc = (time.perf_counter() - t0) / reps
        curve.append({"n": ncv, "ms_per_blocking_call": round(dtc * 1e3, 4), "verifications_per_s": round(ncv / dtc, 1)})
        cc.close()
    out["latency_curve"] = curve
    del dcurve
    # a 64-set batch (the size of one beacon block's signature sets): latency
    d64 = sign_records(m, cache, dev, range(64))
    c64 = m.BatchedBLSVerifierCache.init(max_sets=64, device=dev.index or 0)
    assert c64.verify_device(d64.data_ptr(), 64, rnd)
    t0 = time.perf_counter()
    for _ in range(5):
        assert c64.verify_device(d64.data_ptr(), 64, rnd)
    out["batchVerify_64"] = {"ms_per_blocking_call": (time.perf_counter() - t0) / 5 * 1e3}
    c64.close()
    # config 2: 4 096-tuple batches (latency-bound: the kernels of one such batch fill a sixteenth of the chip)
    n4 = 4096
    d4 = sign_records(m, cache, dev, range(n4))
    NF = 16
    c4 = [m.BatchedBLSVerifierCache.init(max_sets=n4, device=dev.index or 0) for _ in range(NF)]
    s4 = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    for c in c4[1:]:
        c.set_cooperative(False)                   # many batches in flight: one lane per set is the efficient form (and such a context creates no fork streams)
    for c, st in zip(c4, s4):
        assert c.verify_device(d4.data_ptr(), n4, rnd, st.cuda_stream)
    # one blocking caller: on the default stream, as a plain blocking call from C or Nim is.  (On one of the sixteen torch streams created above the
    # caller's main and fork streams can share one of the 8 hardware queues, and its forked kernels then serialise: 6.36 against 5.5 ms measured -
    # an artefact of this many-callers set-up, not of a single caller.)
    torch.cuda.synchronize()
    for _ in range(2):
        assert c4[0].verify_device(d4.data_ptr(), n4, rnd, 0)
    t0 = time.perf_counter()
    for _ in range(5):
        assert c4[0].verify_device(d4.data_ptr(), n4, rnd, 0)
    one4 = (time.perf_counter() - t0) / 5
    c4[0].set_cooperative(False)
    row = {"ms_per_blocking_call": one4 * 1e3, "verifications_per_s_one_caller": n4 / one4}
    for nf in (8, NF):                             # independent callers, one context + stream each
        reps = 9 * nf
        for i in range(nf):                        # warm-up of the other kernel variants
            c4[i].submit_device(d4.data_ptr(), n4, rnd, s4[i].cuda_stream)
        for i in range(nf):
            assert c4[i].wait()
        t0 = time.perf_counter()
        for i in range(reps):
            if i >= nf:
                assert c4[i % nf].wait()
            c4[i % nf].submit_device(d4.data_ptr(), n4, rnd, s4[i % nf].cuda_stream)
        for i in range(nf):
            assert c4[(reps + i) % nf].wait()
        row["verifications_per_s_%d_in_flight" % nf] = n4 / ((time.perf_counter() - t0) / reps)
    row["in_flight_note"] = ("one host thread submits and waits round robin, i.e. in lockstep with the slowest caller; HIP attaches a stream to the least-used of the "
                             "GPU_MAX_HW_QUEUES hardware queues at its first use and callers that share a queue run in turn, so the same code reports ~2.6 M/s when the "
                             "callers' streams fall two per queue and ~1.9 M/s when some queue holds three - which streams the PROCESS used before decides "
                             "(profiles/r06_ab/fork_streams.txt, tests/gpu_probe_hq.py: 2.6 M/s in a fresh process); mi355_bls_batch_verify_many is the entry point for many small batches")
    out["batchVerify_4096"] = row
    for c in c4:
        c.close()
    # the same small batches, sixteen at a time in ONE device pass (mi355_bls_batch_verify_many: every tuple keeps its own batch's
    # blinding scalars, one merged check, per-batch verdicts): small batches at whole-chip throughput, one caller
    km = 16
    dm = sign_records(m, cache, dev, range(1 << 22, (1 << 22) + n4 * km))
    rnds = [hashlib.sha256(b"many" + bytes([i])).digest() for i in range(km)]
    assert m.batchVerifyMany_device(cache, dm.data_ptr(), [n4] * km, rnds) == [True] * km
    t0 = time.perf_counter()
    for _ in range(5):
        assert all(m.batchVerifyMany_device(cache, dm.data_ptr(), [n4] * km, rnds))
    dtm = (time.perf_counter() - t0) / 5
    out["batchVerifyMany_16x4096"] = {"ms_per_call": dtm * 1e3, "verifications_per_s_one_caller": n4 * km / dtm,
                                      "note": "16 independent 4 096-tuple batches per call, per-batch verdicts; one blocking caller of a latency-mode context"}
    nm = 1 << 20
    rng = random.Random(7)
    base = sign_records(m, cache, dev, range(2048), sks=[rng.getrandbits(96) | 1 for _ in range(2048)], msgs=[msg] * 2048)
    dp = base.view(2048, 320)[:, :96].contiguous().repeat(nm // 2048, 1).reshape(-1)     # P_i = [a_i]G1, a_i 96-bit
    sc = np.random.default_rng(7).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    t0 = time.perf_counter()
    for _ in range(5):
        m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    dt = (time.perf_counter() - t0) / 5
    tm = cache.timings()
    # two MSMs in flight (two contexts, two streams, the results left in device memory): what a caller that pipelines its calls gets -
    # the latency-bound reduction of one MSM runs beside the bucket accumulation of the other
    c2 = [m.BatchedBLSVerifierCache.init(max_sets=64, device=dev.index or 0) for _ in range(2)]
    s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
    o2 = [torch.zeros(144, dtype=torch.uint8, device=dev) for _ in range(2)]
    for reps in (2, 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            m.p1s_mult_pippenger_partial_device(c2[i % 2], o2[i % 2].data_ptr(), dp.data_ptr(), nm, ds.data_ptr(), 255, s2[i % 2].cuda_stream)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / reps
    one = m.p1s_mult_pippenger_device(cache, dp.data_ptr(), nm, ds.data_ptr(), 255)
    # the same POINT as the blocking call (the Jacobian image may differ: the counting sort places a bucket's points in atomic order)
    assert g1_jacobian_image_to_affine(bytes(o2[0].cpu().numpy())) == g1_jacobian_image_to_affine(one) == g1_jacobian_image_to_affine(bytes(o2[1].cpu().numpy()))
    for c in c2:
        c.close()
    out["g1_msm_2^20"] = {"points_per_s": nm / dt, "ms_per_call": dt * 1e3, "nbits": 255, "points_per_s_two_in_flight": nm / dt2, "ms_per_msm_two_in_flight": dt2 * 1e3,
                          "GBs_at_128B_per_point": 128.0 * nm / dt / 1e9, "hbm_frac_at_128B_per_point": 128.0 * nm / dt / 1e9 / HBM_PEAK_GBS,
                          "stage_ms": {"sort": tm["blinding"], "buckets": tm["hash_to_g2"], "segments": tm["pk_mul"], "windows": tm["sig_mul_sum"]}}
    return out


# What one BLST core does per signature inside a batch verification, for the crossover ESTIMATE against the real reference (no BLST on the
# box): the reference's README quotes about 1 ms for a single verification and roughly a third of that per signature in a batch (shared
# final exponentiation, N_MAX = 8 Miller batches) on a modern x86 core; the C restatement measured here is several times slower than that.
BLST_US_PER_SIG_IN_BATCH = 400.0
BLST_US_FIXED = 600.0


def crossover(gpu_curve, cpu):
    """Joins the GPU latency curve with the CPU one: per size the blocking-call time of each, the smallest size from which the GPU call is
    faster than (a) the C restatement on this box's cores (MEASURED) and (b) a BLST model on the same number of cores (ESTIMATE:
    BLST_US_FIXED + n * BLST_US_PER_SIG_IN_BATCH / min(n, cores); BLST itself is not on the box)."""
    cores = cpu.get("cores", 1)
    ccurve = {r["n"]: r for r in cpu.get("latency_curve", [])}
    rows, n_port, n_blst = [], None, None
    for g in gpu_curve:
        n = g["n"]
        blst_ms = (BLST_US_FIXED + n * BLST_US_PER_SIG_IN_BATCH / max(1, min(n, cores))) / 1e3
        row = {"n": n, "gpu_ms": g["ms_per_blocking_call"], "cpu_port_ms": ccurve.get(n, {}).get("ms_per_call"), "blst_model_ms": round(blst_ms, 3)}
        rows.append(row)
    for r in rows:                                   # the first size from which the GPU stays ahead
        if n_port is None and r["cpu_port_ms"] is not None and all(q["gpu_ms"] < q["cpu_port_ms"] for q in rows if q["n"] >= r["n"] and q["cpu_port_ms"] is not None):
            n_port = r["n"]
        if n_blst is None and all(q["gpu_ms"] < q["blst_model_ms"] for q in rows if q["n"] >= r["n"]):
            n_blst = r["n"]
    return {"rows": rows, "cores": cores, "gpu_faster_than_cpu_port_from_n": n_port, "gpu_faster_than_blst_model_from_n": n_blst,
            "blst_model": "ESTIMATE, not a measurement: %.0f us + n x %.0f us / min(n, cores) - BLST is not installed on the box" % (BLST_US_FIXED, BLST_US_PER_SIG_IN_BATCH),
            "note": "one blocking batchVerify per size on each side; below the crossover a host should keep its CPU path (INTEGRATION.md, 'When to call the GPU')"}


def host_cores():
    """Cores this process may use: the scheduler affinity mask capped by the cgroup CPU quota (a container often sees all
    of the host's CPUs in the mask while being allowed a fraction of them)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    cores = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return cores, aff, quota


def blst_baseline(recs, rnd, threads):
    """If the box has a libblst: the reference's own scheme on it (one blst_pairing per thread, parallel_chunks split,
    blst_pairing_chk_n_mul_n_aggr_pk_in_g1 per tuple, commit, merge, finalverify; blst_abi.nim:462-509,
    bls_batch_verifier.nim:326-371), threads = Python threads (ctypes releases the GIL).  None when there is no libblst."""
    path = ctypes.util.find_library("blst")
    if not path:
        return None
    try:
        B = ctypes.CDLL(path)
        vp, sz, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p
        B.blst_pairing_sizeof.restype = sz
        B.blst_pairing_init.argtypes = [vp, ctypes.c_int, cp, sz]
        B.blst_pairing_chk_n_mul_n_aggr_pk_in_g1.argtypes = [vp, vp, ctypes.c_int, vp, ctypes.c_int, cp, sz, cp, sz, cp, sz]
        B.blst_pairing_commit.argtypes = [vp]
        B.blst_pairing_merge.argtypes = [vp, vp]
        B.blst_pairing_finalverify.argtypes = [vp, vp]
        dst = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"
        n = len(recs) // 320
        T = min(threads, n)
        base, rem = divmod(n, T)
        ctxs = [ctypes.create_string_buffer(B.blst_pairing_sizeof()) for _ in range(T)]
        buf = ctypes.create_string_buffer(recs, len(recs))
        addr = ctypes.addressof(buf)

        def chunk(c):
            off = (base + 1) * c if c < rem else base * c + rem
            ln = base + 1 if c < rem else base
            B.blst_pairing_init(ctxs[c], 1, dst, len(dst))
            seed = hashlib.sha256(rnd + c.to_bytes(8, "little")).digest()
            for i in range(off, off + ln):
                while True:
                    seed = hashlib.sha256(seed).digest()
                    if seed[:8] != bytes(8):
                        break
                rc = B.blst_pairing_chk_n_mul_n_aggr_pk_in_g1(ctxs[c], addr + 320 * i, 0, addr + 320 * i + 128, 0, seed[:8], 64,
                                                              recs[320 * i + 96:320 * i + 128], 32, None, 0)
                if rc != 0:
                    return False
            B.blst_pairing_commit(ctxs[c])
            return True
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=T) as pool:
            ok = all(pool.map(chunk, range(T)))
        for c in range(1, T):
            B.blst_pairing_merge(ctxs[0], ctxs[c])
        ok = ok and bool(B.blst_pairing_finalverify(ctxs[0], None))
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "verifications/s", "cores": T, "verdict": ok, "lib": path}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e), "lib": path}


def cpu_baseline(co, sample, rnd):
    """The C restatement of the reference algorithm (one pairing context per thread, parallel_chunks
    split, update loop, commit, merge, finalverify) on the host cores, the whole leg bounded to roughly 30 s."""
    cores, aff, quota = host_cores()
    probe = co.make_batch(max(16 * cores, 64), seed=1 << 40)
    co.set_num_threads(1)
    assert co.batch_verify(probe[:320 * 8], rnd, 1)          # warm-up: the first call pays for loading the library and its tables
    # 1 -> N thread scaling on >= 64 tuples (16 per thread: two full N_MAX = 8 Miller batches each), which tells a real core
    # count from an oversubscribed one
    scaling = []
    t = 1
    while True:
        co.set_num_threads(t)
        k = max(64, 16 * t)
        t0 = time.perf_counter()
        assert co.batch_verify(probe[:320 * k], rnd, t)
        scaling.append({"threads": t, "tuples": k, "verifications_per_s": round(k / (time.perf_counter() - t0), 1)})
        if t >= cores:
            break
        t = min(cores, t * 2)
    best = max(scaling, key=lambda s: s["verifications_per_s"])
    use = best["threads"]
    co.set_num_threads(use)
    if sample <= 0:
        sample = int(max(2 * use, min(65536, 10.0 * best["verifications_per_s"])))
    recs = co.make_batch(min(sample, 4096), seed=(1 << 40) + 7)
    if sample > 4096:
        recs = (recs * ((sample + 4095) // 4096))[:320 * sample]
    t0 = time.perf_counter()
    ok = co.batch_verify(recs, rnd, use)
    dt = time.perf_counter() - t0
    assert ok
    out = {"value": sample / dt, "unit": "verifications/s", "cores": use, "kind": "port",
           "value_1core": scaling[0]["verifications_per_s"], "thread_scaling": scaling,
           "host": {"sched_affinity_cpus": aff, "cgroup_cpu_quota": quota, "threads_used": use},
           "sample": "%d tuples of the same workload, batchVerifyParallel shape with %d threads, %.1f s (oracle/bls_oracle.c: plain-C restatement of "
                     "the reference algorithm, not BLST; BLST's assembly is several times faster per core)" % (sample, use, dt)}
    # the same sizes as aux.latency_curve, one blocking call each on min(n, cores) threads (the reference's B = min(n, numThreads)); the
    # headline size is the main leg above
    ccurve = []
    for ncv in LATENCY_CURVE_SIZES:
        if ncv > 16384:
            ccurve.append({"n": ncv, "ms_per_call": round(ncv / (sample / dt) * 1e3, 3), "verifications_per_s": round(sample / dt, 1), "threads": use, "from": "the main leg's rate"})
            continue
        tcv = max(1, min(ncv, use))
        co.set_num_threads(tcv)
        rcv = (recs * ((ncv + sample - 1) // sample))[:320 * ncv] if ncv > sample else recs[:320 * ncv]
        reps = 3 if ncv <= 1024 else 1
        assert co.batch_verify(rcv, rnd, tcv)
        t0 = time.perf_counter()
        for _ in range(reps):
            assert co.batch_verify(rcv, rnd, tcv)
        dcv = (time.perf_counter() - t0) / reps
        ccurve.append({"n": ncv, "ms_per_call": round(dcv * 1e3, 3), "verifications_per_s": round(ncv / dcv, 1), "threads": tcv})
    co.set_num_threads(use)
    out["latency_curve"] = ccurve
    blst = blst_baseline(recs[:320 * min(sample, 16384)], rnd, use)
    out["blst_on_this_box"] = blst is not None
    if blst is not None:
        out["blst"] = blst
        if "value" in blst:
            out.update({"value": blst["value"], "kind": "reference", "cores": blst["cores"], "port_value": sample / dt})
    # ---- legs for the other BASELINE configs (bounded samples; 'port' like the main leg)
    legs = {}
    msg = b"Mr F was here"
    sk = 0x263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3
    pk = co.sk_to_pk(sk)
    t0 = time.perf_counter()
    for _ in range(20):
        sig = co.sign(sk, msg)
    ts = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        assert co.core_verify(pk, msg, sig)
    tv = (time.perf_counter() - t0) / 20
    legs["config1_sign_verify"] = {"sign_ops_per_s": 1 / ts, "verify_ops_per_s": 1 / tv, "cores": 1,
                                   "sample": "benchmarks/bls_signature.nim:115-143 shape: msg 'Mr F was here', 20 sign + 20 verify on one core"}
    n4 = min(4096, max(64, int(2.0 * best["verifications_per_s"])))
    t0 = time.perf_counter()
    assert co.batch_verify(recs[:320 * n4], rnd, use)
    legs["config2_batch_4096"] = {"verifications_per_s": n4 / (time.perf_counter() - t0), "cores": use, "sample": "%d of the 4096 tuples" % n4}
    nk = 32768
    pks, sksum = co.make_pks(4096, seed=1 << 41)
    pks = pks * (nk // 4096)
    h = co.hash_to_g2(msg, b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_")
    fsig = co.g2_mul(h, sksum * (nk // 4096) % R_ORDER)
    t0 = time.perf_counter()
    assert co.fast_aggregate_verify(pks, msg, fsig)
    legs["config3_fastAggregateVerify_32768"] = {"ms_per_call": (time.perf_counter() - t0) * 1e3, "cores": 1,
                                                 "sample": "one call, 32 768 keys (4096 distinct, tiled), serial G1 sum as aggregateAll does (core :179-195)"}
    nm = 1 << 16
    import numpy as np
    sc = np.random.default_rng(11).integers(0, 256, size=(nm, 32), dtype=np.uint8).tobytes()
    mp = (pks[:96 * 4096] * (nm // 4096))
    t0 = time.perf_counter()
    co.msm_g1_pippenger(mp, sc, 255)
    dtm = time.perf_counter() - t0
    legs["config4_g1_msm"] = {"points_per_s": nm / dtm, "cores": use, "sample": "2^16 of the 2^20 points, nbits 255, Pippenger restatement with windows on OpenMP threads, %.1f s" % dtm}
    # row f1: fromBytes of every tuple (decompression, not-infinity, subgroup checks), the restatement's OpenMP loop over the tuples
    nf = 64 * use
    recs = co.make_batch(nf, seed=1 << 41)
    pk48, sg96 = compress_records(recs)
    ms = b"".join(recs[320 * i + 96:320 * i + 128] for i in range(nf))
    t0 = time.perf_counter()
    okd, back, _ = co.deserialize_sets(pk48, ms, sg96)
    dtf = time.perf_counter() - t0
    assert okd and back == recs
    legs["f1_fromBytes"] = {"tuples_per_s": nf / dtf, "cores": use,
                            "sample": "%d tuples; the restatement tests subgroup membership as [r]P = O (BLST uses the endomorphism tests, several times cheaper)" % nf}
    out["legs"] = legs
    return out


if __name__ == "__main__":
    main()
