"""Ad-hoc GPU probe: batched deserialisation + validation timing (not a test)."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
m = ge.load_package()
import c_oracle as co
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rec = co.make_batch(n, seed=5)
pks, msgs, sigs = co.compress_sets(rec)
cache = m.BatchedBLSVerifierCache.init(max_sets=n)
rnd = hashlib.sha256(b"Mr F was here").digest()
for it in range(3):
    t0 = time.time(); ok, st = m.batchVerifyCompressed(cache, pks, msgs, sigs, rnd); dt = time.time() - t0
    print(n, ok, "wall %.1f ms" % (dt * 1e3), "deser kernel %.2f ms" % m.lib().mi355_bls_last_deser_ms(cache._h), cache.timings()["total"], flush=True)
