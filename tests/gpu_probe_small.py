"""Ad-hoc GPU probe (not a test): throughput of many 4 096-tuple batches in flight (BASELINE config 2 shape) against the number of callers
and of HIP hardware queues.  usage: GPU_MAX_HW_QUEUES=q python3 tests/gpu_probe_small.py [nf ...]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
n4 = int(os.environ.get("PROBE_N", "4096"))
gen = m.BatchedBLSVerifierCache.init(max_sets=n4)
d4 = bench.sign_records(m, gen, dev, range(n4))
for nf in [int(x) for x in sys.argv[1:]] or [8, 16, 32]:
    s4 = [torch.cuda.Stream(device=dev) for _ in range(nf)]
    c4 = [m.BatchedBLSVerifierCache.init(max_sets=n4) for _ in range(nf)]
    for c in c4:
        c.set_cooperative(os.environ.get("PROBE_COOP", "0") == "1")
    for i in range(nf):
        c4[i].submit_device(d4.data_ptr(), n4, rnd, s4[i].cuda_stream)
    for i in range(nf):
        assert c4[i].wait()
    reps = 12 * nf
    t0 = time.perf_counter()
    for i in range(reps):
        if i >= nf:
            assert c4[i % nf].wait()
        c4[i % nf].submit_device(d4.data_ptr(), n4, rnd, s4[i % nf].cuda_stream)
    for i in range(nf):
        assert c4[(reps + i) % nf].wait()
    dt = (time.perf_counter() - t0) / reps
    print("queues", os.environ.get("GPU_MAX_HW_QUEUES", "default"), "n", n4, "in flight", nf, "coop", os.environ.get("PROBE_COOP", "0"), "-> %.2f M verifications/s" % (n4 / dt / 1e6), flush=True)
    for c in c4:
        c.close()
