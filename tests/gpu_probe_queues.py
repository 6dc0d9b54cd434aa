"""Probe: which torch.cuda streams can run kernels concurrently (HW-queue sharing)."""
import time, sys, os
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream() for _ in range(n)]
cyc = 200_000_000  # ~0.1 s
def run(idx):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx:
        with torch.cuda.stream(streams[i]):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run([0]); run([1])
base = min(run([0]) for _ in range(3))
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "single %.3f" % base)
for i in range(n):
    row = []
    for j in range(n):
        row.append("." if i == j else ("S" if run([i, j]) > 1.5 * base else "c"))
    print(i, " ".join(row))
print("all %d: %.2fx" % (n, run(list(range(n))) / base))
