"""The N > 1 bench path (sharding by chunk blocks, 584-byte exchange, merge + final exponentiation on rank 0)
end to end on ONE GPU: two ranks share device 0 and exchange over gloo (bench.py's documented test hook);
real multi-GPU runs use RCCL with one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("batch,port", [(4096, 29533), (65536, 29534)])
def test_two_ranks_on_one_gpu(batch, port):
    """batch = 65536: the per-GPU shard size of the headline configuration, every step's merged verdict asserted by the bench."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_ALL_ON_DEVICE0="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", str(batch),
           "--no-cpu", "--no-aux"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * batch and d["value"] > 0 and d["scaling"] == "weak"


def test_rccl_exchange_with_one_rank():
    """The device-resident exchange of the N > 1 path (shard blob -> RCCL all_gather on device buffers -> merge + final
    exponentiation, verdict read one step later) through a real RCCL communicator of one rank (an 8-GPU node is not available to
    the tests): `bench.py --force-dist`."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "4096", "--no-cpu", "--no-aux", "--force-dist"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["exchange"] == {"mode": "device"} and d["value"] > 0
    assert d["dist"]["backend"] == "nccl" and d["dist"]["is_rccl"] and d["dist"]["world_size"] == 1 and d["dist"]["nccl_version"]


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes itself (before any GPU call of its own),
    relays rank 0's JSON line and returns 0; both ranks share device 0 and exchange over gloo (the documented test hook).  The
    multi-GPU MSM rows ride along."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_ALL_ON_DEVICE0="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4096", "--no-cpu"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["exchange"]["mode"] == "host" and d["value"] > 0
    msm = d["multi_gpu"]["g1_msm"]
    assert msm["weak_2^20_per_gpu"]["points_total"] == 2 << 20 and msm["weak_2^20_per_gpu"]["points_per_s"] > 0
    assert msm["strong_2^20_total"]["points_total"] == 1 << 20


def test_eight_ranks_dry_run_on_one_gpu():
    """The `--gpus 8` command the driver runs on an 8-GPU node, as a dry run on ONE device (eight rank processes share device 0 and
    exchange over gloo; 4 096 tuples per rank): the headline loop, the multi-GPU MSM rows and the config-5 row (twice the batch per
    GPU, the shape of BASELINE config 5) all execute, so that the first real 8-GPU run is not also the first run of this code path.
    No scaling number is read from this."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_ALL_ON_DEVICE0="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "4096", "--no-cpu", "--msm-log2", "14"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * 4096 and d["value"] > 0
    c5 = d["multi_gpu"]["config5_shape_batchVerify"]
    assert c5["tuples_per_gpu"] == 8192 and c5["global_batch"] == 8 * 8192 and c5["verifications_per_s"] > 0
    assert d["multi_gpu"]["g1_msm"]["weak_2^14_per_gpu"]["points_total"] == 8 << 14
    # the strong-scaling row beside the weak headline: one --batch-sized batch in all, batch / 8 per GPU
    st = d["multi_gpu"]["batchVerify_strong"]
    assert st["scaling"] == "strong" and st["global_batch"] == 4096 and st["tuples_per_gpu"] == 512 and st["verifications_per_s"] > 0
    assert d["scaling"] == "weak" and d["config"]["multi_batchVerify_strong_ms_per_step"] == st["ms_per_step"]      # repeated flat for the driver's record
    # what stands behind the N > 1 line (round-4 review): backend, world size and one device identity per rank, over the control group
    di = d["dist"]
    assert di["backend"] == "gloo" and di["is_rccl"] is False and di["world_size"] == 8 and len(di["device_uuids"]) == 8
    assert di["distinct_devices"] == 1 and di["all_on_device0_test_hook"] is True        # the dry run: eight ranks on ONE device, and the line says so
    assert d["build"]["aligned"] is True and d["build"]["dpp_combine_off"] is True and len(d["build"]["stamp"]) == 64
    rf = d["roofline"]
    assert rf["bound"] == "int_mad" and rf["algorithmic_bytes_per_unit"] == 320 and rf["units_per_launch"] == 4096
    assert abs(rf["frac"] - 320 * 4096 / (d["ms_per_step"] * 1e-3) / 8e12) < 1e-9
    assert set(rf["kernels"]) == {"k_hash_map", "k_hash_clear", "k_pkmul", "k_sig_bucket", "k_lines", "k_lineprod"}
    assert rf["int_mad"]["ceiling"]["measured_on_this_box"] is True and 0.5 < rf["int_mad"]["ceiling"]["frac_of_peak"] < 1.0


def test_ranks_on_one_device_without_the_hook_fail_loudly():
    """Two ranks that land on ONE device without BENCH_ALL_ON_DEVICE0 (what a mis-set LOCAL_RANK / visible-devices mask would do on a real
    node): the run must refuse instead of reporting the scaling of one GPU as that of two."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    env.pop("BENCH_ALL_ON_DEVICE0", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["BENCH_FORCE_LOCAL0"] = "1"          # test hook: every rank uses device 0 WITHOUT declaring the dry run
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "64", "--no-cpu", "--no-aux"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "distinct devices" in p.stderr, p.stderr[-2000:]


def test_a_dying_rank_fails_the_run():
    """One of the rank processes `bench.py --gpus 2` started dies after its set-up (test hook BENCH_TEST_DIE_RANK): the parent must end
    the other rank, report which rank failed and exit non-zero well within its timeout - not hang in the first exchange."""
    import time
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_ALL_ON_DEVICE0="1", BENCH_TEST_DIE_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096", "--no-cpu", "--no-aux"]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 1, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with code 3" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]            # no result line from a failed run
    assert time.time() - t0 < 300


def test_bench_refuses_a_world_that_does_not_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
