"""Ad-hoc GPU probe: throughput with 1, 2, 3 concurrent callers (one context + stream each)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import __graft_entry__ as ge
m = ge.load_package()
import c_oracle as co
import hashlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rec = co.make_batch(8192, seed=3)
rec = (rec * (n // 8192 + 1))[:320 * n]
rnd = hashlib.sha256(b"Mr F was here").digest()
d = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
for nthreads in (1, 2, 3):
    caches = [m.BatchedBLSVerifierCache.init(max_sets=n) for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    steps = 6 * nthreads
    def worker(t):
        for i in range(steps // nthreads):
            assert caches[t].verify_device(d.data_ptr(), n, rnd, streams[t].cuda_stream)
    for t in range(nthreads):
        caches[t].verify_device(d.data_ptr(), n, rnd, streams[t].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("callers", nthreads, "steps", steps, "ms/step %.2f" % (dt / steps * 1e3), "verif/s %.0f" % (n * steps / dt), flush=True)
    for c in caches: c.close()
