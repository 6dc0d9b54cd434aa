"""Ad-hoc GPU soak (the long form; tests/test_gpu_soak.py runs a fixed-seed slice of it under pytest -m gpu): randomised differential run of the batch entry points against the C restatement -
random batch sizes, context capacities (slices), chain counts, context modes, host / device records, valid and tampered batches, and
the many-batches entry point.  usage: python3 tests/gpu_soak.py [iterations] [seed]"""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import __graft_entry__ as ge
import c_oracle as co


def soak(iters, seed, m=None, verbose=True):
    """`iters` randomised iterations from `seed`; raises AssertionError with the failing configuration; returns the iteration count"""
    if m is None:
        m = ge.load_package()
    rng = random.Random(seed)
    pool = co.make_batch(6000, seed=424243)                       # valid tuples to draw batches from
    t0 = time.time()
    for it in range(iters):
        n = rng.choice([1, 2, 3, rng.randrange(4, 70), rng.randrange(70, 3000)])
        start = rng.randrange(0, 6000 - n)
        rec = bytearray(pool[320 * start:320 * (start + n)])
        kind = rng.choice(["valid", "valid", "msg", "swap", "infpk", "infsig"])
        if kind == "msg":
            rec[320 * rng.randrange(n) + 96 + rng.randrange(32)] ^= 1 << rng.randrange(8)
        elif kind == "swap" and n >= 2:
            i, j = rng.sample(range(n), 2)
            rec[320 * i + 128:320 * i + 320], rec[320 * j + 128:320 * j + 320] = rec[320 * j + 128:320 * j + 320], rec[320 * i + 128:320 * i + 320]
        elif kind == "infpk":
            i = rng.randrange(n)
            rec[320 * i:320 * i + 96] = bytes(96)
        elif kind == "infsig":
            i = rng.randrange(n)
            rec[320 * i + 128:320 * i + 320] = bytes(192)
        rec = bytes(rec)
        rnd = hashlib.sha256(b"soak" + it.to_bytes(4, "little")).digest()
        nt = rng.choice([1, 2, 4, 7, 64, 333, 4096])
        cap = rng.choice([1, 2, 5, 64, 100, 777, 4096])
        serial = nt == 1 or n < 3 or rng.random() < 0.15
        want, st = co.batch_verify(rec, rnd, 0 if serial else nt, stages=True)
        cache = m.BatchedBLSVerifierCache.init(max_sets=cap, numThreads=nt)
        cache.set_cooperative(rng.random() < 0.5)
        how = rng.choice(["host", "device", "many"])
        if serial and (nt > 1 and n >= 3):
            got = m.batchVerifySerial(cache, rec, rnd)
            how = "serial"
        elif how == "host":
            got = m.batchVerify(cache, rec, rnd)
        elif how == "device":
            d = torch.frombuffer(bytearray(rec), dtype=torch.uint8).cuda()
            if nt > 1 and n >= 3:
                cache.submit_device(d.data_ptr(), n, rnd)
                got = cache.wait()
            else:
                got = m.batchVerify(cache, rec, rnd)
        else:
            # the batch cut into 1..4 consecutive sub-batches, each with its own random bytes: verdicts per sub-batch
            k = rng.randrange(1, 5)
            cuts = sorted(rng.sample(range(1, n), min(k - 1, n - 1))) if n > 1 else []
            parts = [rec[320 * a:320 * b] for a, b in zip([0] + cuts, cuts + [n])]
            rnds = [hashlib.sha256(rnd + bytes([j])).digest() for j in range(len(parts))]
            gotm = m.batchVerifyMany(cache, parts, rnds)
            wantm = [co.batch_verify(p, r, nt if (nt > 1 and len(p) // 320 >= 3) else 0) for p, r in zip(parts, rnds)]
            assert gotm == wantm, (it, "many", n, nt, cap, kind, gotm, wantm)
            cache.close()
            continue
        assert got == want, (it, how, n, nt, cap, kind, got, want)
        # the GT value is comparable whenever every point is on its curve and no key is at infinity (then the reference stops early)
        if kind != "infpk":
            assert cache.fetch(4, 576) == st["gt"], (it, how, n, nt, cap, kind, "gt")
        cache.close()
        if it % 25 == 24:
            if verbose: print("soak: %d iterations ok, %.0f s" % (it + 1, time.time() - t0), flush=True)

    return iters


if __name__ == "__main__":
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    soak(n_it, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("soak: all %d iterations ok" % n_it)
