"""Ad-hoc GPU probe: stage timings at several batch sizes (not a test)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
m = ge.load_package()
g = json.load(open(os.path.join(ROOT, "tests", "golden", "batch.json")))
c = [x for x in g["cases"] if x["name"] == "n17"][0]
rec, rnd = bytes.fromhex(c["sets"]), bytes.fromhex(c["rnd"])
recs = [rec[320 * i:320 * i + 320] for i in range(17)]
for n in [int(a) for a in sys.argv[1:]] or [64, 1024, 4096]:
    big = b"".join(recs[i % 17] for i in range(n))
    cache = m.BatchedBLSVerifierCache.init(max_sets=n)
    for it in range(2):
        t0 = time.time()
        ok = m.batchVerify(cache, big, rnd)
        dt = time.time() - t0
        print(n, ok, "wall %.1f ms" % (dt * 1e3), {k: round(v, 3) for k, v in cache.timings().items()}, flush=True)
    cache.close()
