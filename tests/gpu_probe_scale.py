import hashlib, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge
import bench
m = ge.load_package()
dev = torch.device("cuda", 0)
rnd = hashlib.sha256(b"Mr F was here").digest()
gen = m.BatchedBLSVerifierCache.init(max_sets=65536)
d = torch.cat([bench.sign_records(m, gen, dev, range(s, s + 65536)) for s in (0, 65536)])
for n in (8192, 16384, 32768, 49152, 65536, 98304, 131072):
    c = m.BatchedBLSVerifierCache.init(max_sets=n)
    c.set_cooperative(False)
    acc = {}
    for _ in range(4):
        assert c.verify_device(d.data_ptr(), n, rnd)
        for k, v in list(c.kernel_timings().items()) + list(c.timings().items()):
            acc[k] = acc.get(k, 0) + v / 4
    print(n, {k: round(v, 3) for k, v in acc.items() if k in ("k_hash_map", "k_hash_clear", "k_lineprod", "pk_mul", "miller_lines")})
    c.close()
