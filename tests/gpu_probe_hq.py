import hashlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
n4 = 4096
gen = m.BatchedBLSVerifierCache.init(max_sets=n4)
d4 = bench.sign_records(m, gen, dev, range(n4))
npre = int(sys.argv[1])
pre = [m.BatchedBLSVerifierCache.init(max_sets=64) for _ in range(npre)]
for c in pre:
    assert c.verify_device(d4.data_ptr(), 64, rnd)          # a forked latency-mode call: the context's fork streams get their hardware queues
for nf in (8, 16):
    s4 = [torch.cuda.Stream(device=dev) for _ in range(nf)]
    c4 = [m.BatchedBLSVerifierCache.init(max_sets=n4) for _ in range(nf)]
    for c in c4:
        c.set_cooperative(False)
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
    print("latency contexts that forked before: %d; %d throughput callers in flight -> %.2f M verifications/s" % (npre, nf, n4 / dt / 1e6), flush=True)
    for c in c4:
        c.close()
