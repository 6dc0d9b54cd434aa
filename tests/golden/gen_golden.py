#!/usr/bin/env python3
"""Generates tests/golden/*.json from the KAT-pinned big-integer oracle (oracle/bls12381_py.py).

Run once in the build container:  python tests/golden/gen_golden.py
The fixtures are data only (inputs + expected outputs).  Scenario shapes follow the reference's
tests/t_batch_verifier.nim:34-274 (keyGen(seed), msg = SHA256(text), rnd = SHA256("Mr F was here"))
and benchmarks/bls12381_msm_g1.nim:22-59 (MSM shape); expected verdicts are the ones the reference's
own test asserts.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bls12381_py as o  # noqa: E402

RND = o.sha256(b"Mr F was here")
assert RND.hex().startswith("3e894140")


def hx(b):
    return bytes(b).hex()


_key_cache = {}


def keygen(seed):
    if seed not in _key_cache:
        _key_cache[seed] = o.keygen_seed(seed)
    return _key_cache[seed]


def example(seed, text):
    pk, sk = keygen(seed)
    m = o.sha256(text.encode())
    return (pk, m, o.sign(sk, m))


def forged_pair(seed1, t1, seed2, t2):
    """tests/t_batch_verifier.nim:198-244."""
    pk1, sk1 = keygen(seed1)
    m1 = o.sha256(t1.encode())
    s1 = o.sign(sk1, m1)
    pk2, sk2 = keygen(seed2)
    m2 = o.sha256(t2.encode())
    s2 = o.sign(sk2, m2)
    _, skp = keygen(seed1 * seed2 + seed1 + seed2)
    sp = o.sign(skp, o.sha256(b"rekt"))
    f1 = o.g2_add(s1, sp)
    f2 = o.g2_add(s2, o.g2_neg(sp))
    assert o.aggregate_verify([pk1, pk2], [m1, m2], o.g2_add(f1, f2))
    return [(pk1, m1, f1), (pk2, m2, f2)]


def f12_hex(a):
    return [[hx(o.fp_to_mont_bytes(c[0])), hx(o.fp_to_mont_bytes(c[1]))] for c in o.f12_to_tower_ints(a)]


def batch_case(name, sets, expect, with_stages=True, chunks=(None, 4)):
    case = {
        "name": name,
        "n": len(sets),
        "rnd": hx(RND),
        "sets": hx(b"".join(o.signature_set_bytes(*s) for s in sets)),
        "expect": expect,
    }
    for nc in chunks:
        tag = "serial" if nc is None else "chunks%d" % nc
        st = o.batch_verify_stages(sets, RND, nc)
        assert st["verdict"] == expect, (name, tag, st["verdict"])
        ent = {"r": [str(r) for r in st.get("r", [])]}
        if with_stages and "gt" in st:
            ent["H"] = [hx(o.g2_to_blst_affine(h)) for h in st["H"]]
            ent["rPK"] = [hx(o.g1_to_blst_affine(p)) for p in st["rPK"]]
            ent["aggsig"] = hx(o.g2_to_blst_affine(st["aggsig"]))
            ent["gt"] = f12_hex(st["gt"])
        case[tag] = ent
    print("batch case", name, len(sets), expect, flush=True)
    return case


def gen_batch():
    cases = []
    pk, sk = keygen(123)
    m = o.sha256(b"message")
    cases.append(batch_case("single", [(pk, m, o.sign(sk, m))], True))
    cases.append(batch_case("two", [example(1, "msg1"), example(2, "msg2")], True))
    for n in (3, 8, 9, 15, 16, 17):
        cases.append(batch_case("n%d" % n, [example(i, "msg%d" % i) for i in range(n)], True))
    e1 = example(1, "msg1")
    pk2, _ = keygen(2)
    cases.append(batch_case("wrong_sig", [e1, (pk2, o.sha256(b"msg2"), e1[2])], False))
    cases.append(batch_case("forged_pair", forged_pair(1, "msg1", 2, "msg2"), False))
    many = [example(i, "msg%d" % i) for i in range(16)] + forged_pair(1, "msg100", 2, "msg200")
    random.Random(1234).shuffle(many)
    cases.append(batch_case("forged_among_many", many, False, with_stages=False))
    # infinity public key -> false (BLST_PK_IS_INFINITY), infinity signature is tolerated as input
    good = [example(i, "msg%d" % i) for i in range(3)]
    cases.append(batch_case("inf_pk", [good[0], (None, good[1][1], good[1][2]), good[2]], False,
                            with_stages=False))
    cases.append(batch_case("inf_sig", [good[0], (good[1][0], good[1][1], None), good[2]], False))
    # 100 signatures on the same message (t_batch_verifier.nim:139-177), via batch and via combine
    msg = o.sha256(b"msg")
    pks, sigs = [], []
    hm = o.hash_to_g2(msg)
    for i in range(100):
        pk, sk = keygen(i)
        pks.append(pk)
        sigs.append(o.g2_mul(hm, sk))
    cases.append(batch_case("same_msg_100", list(zip(pks, [msg] * 100, sigs)), True,
                            with_stages=False, chunks=(4,)))
    cpk, csig = o.combine(RND, pks, sigs)
    comb = {
        "rnd": hx(RND), "n": 100, "msg": hx(msg),
        "pks": hx(b"".join(o.g1_to_blst_affine(p) for p in pks)),
        "sigs": hx(b"".join(o.g2_to_blst_affine(s) for s in sigs)),
        "scalars": [str(s) for s in o.combine_scalars(RND, 100)],
        "out_pk": hx(o.g1_to_blst_affine(cpk)), "out_sig": hx(o.g2_to_blst_affine(csig)),
    }
    assert o.batch_verify([(cpk, msg, csig)], RND)
    der = sigs[1:] + sigs[:1]           # fixed derangement instead of the reference's unseeded shuffle
    dpk, dsig = o.combine(RND, pks, der)
    assert not o.batch_verify([(dpk, msg, dsig)], RND)
    comb["deranged_out_sig"] = hx(o.g2_to_blst_affine(dsig))
    return {"cases": cases, "combine": comb}


def gen_fields():
    rng = random.Random(0xFACADE)
    out = {"p": hex(o.P), "fp_mul": [], "fp2_mul": [], "fp_inv": [], "fp2_sqrt": []}
    edge = [0, 1, o.P - 1, o.P - 2, 2, (o.P - 1) // 2, (o.P + 1) // 2]
    vals = edge + [rng.randrange(o.P) for _ in range(25)]
    for a in vals:
        for b in (vals[3], vals[-1], a):
            out["fp_mul"].append([hx(o.fp_to_mont_bytes(a)), hx(o.fp_to_mont_bytes(b)),
                                  hx(o.fp_to_mont_bytes(a * b % o.P)),
                                  hx(o.fp_to_mont_bytes((a + b) % o.P)), hx(o.fp_to_mont_bytes((a - b) % o.P))])
        if a:
            out["fp_inv"].append([hx(o.fp_to_mont_bytes(a)), hx(o.fp_to_mont_bytes(o.fp_inv(a)))])
    for _ in range(16):
        a = (rng.randrange(o.P), rng.randrange(o.P))
        b = (rng.randrange(o.P), rng.randrange(o.P))
        c = o.f2mul(a, b)
        s = o.f2sqr(a)
        out["fp2_mul"].append([[hx(o.fp_to_mont_bytes(v)) for v in t] for t in (a, b, c, s, o.f2inv(a))])
    return out


def gen_h2c():
    out = []
    msgs = [b"", b"abc", o.sha256(b"msg0"), o.sha256(b"message"), bytes(range(32)), b"\xff" * 32]
    for m in msgs:
        for dst in (o.DST_SIG, o.DST_POP):
            u = o.hash_to_field_fp2(m, dst)
            q0 = o.sswu_g2(u[0])
            q1 = o.sswu_g2(u[1])
            h = o.hash_to_g2(m, dst)
            out.append({
                "msg": hx(m), "dst": dst.decode(),
                "xmd": hx(o.expand_message_xmd(m, dst, 256)),
                "u": [[hx(o.fp_to_mont_bytes(c)) for c in e] for e in u],
                "q0": hx(o.g2_to_blst_affine(q0)), "q1": hx(o.g2_to_blst_affine(q1)),
                "h": hx(o.g2_to_blst_affine(h)), "h_compressed": hx(o.g2_compress(h)),
            })
    return out


def gen_pairing():
    rng = random.Random(7)
    out = []
    for _ in range(4):
        a = rng.randrange(1, o.R)
        b = rng.randrange(1, o.R)
        p = o.g1_mul(o.G1_GEN, a)
        q = o.g2_mul(o.G2_GEN, b)
        out.append({"p": hx(o.g1_to_blst_affine(p)), "q": hx(o.g2_to_blst_affine(q)),
                    "gt3": f12_hex(o.pairing(p, q))})
    return {"note": "gt3 = e(P,Q)^3 = miller^(3*(p^12-1)/r) (HHT hard part)", "vectors": out}


def gen_msm():
    """benchmarks/bls12381_msm_g1.nim:22-59 shape: P_i=[a_i]G (96-bit a_i), 32-byte scalars, nbits=255."""
    rng = random.Random(0xFACADE)
    out = []
    for n in (1, 2, 31, 32, 33, 256):
        pts = [o.g1_mul(o.G1_GEN, rng.getrandbits(96) | 1) for _ in range(n)]
        sc = [rng.getrandbits(256) for _ in range(n)]
        res = o.msm_g1(pts, sc, 255)
        out.append({"n": n, "nbits": 255,
                    "points": hx(b"".join(o.g1_to_blst_affine(p) for p in pts)),
                    "scalars": hx(b"".join(s.to_bytes(32, "little") for s in sc)),
                    "result_affine": hx(o.g1_to_blst_affine(res))})
        print("msm", n, flush=True)
    # aggregate (fastAggregateVerify's G1 sum)
    pts = [o.g1_mul(o.G1_GEN, rng.getrandbits(96) | 1) for _ in range(257)]
    agg = {"n": 257, "points": hx(b"".join(o.g1_to_blst_affine(p) for p in pts)),
           "sum_affine": hx(o.g1_to_blst_affine(o.aggregate_g1(pts)))}
    return {"msm": out, "aggregate": agg}


def gen_fav():
    """fastAggregateVerify (bls_sig_min_pubkey.nim:234-258; benchmarks/bls_signature.nim:176-198 shape)."""
    out = []
    msg = b"Mr F was here"
    hm = o.hash_to_g2(msg)
    for n in (1, 2, 128):
        sks = [keygen(i)[1] for i in range(n)]
        pks = [keygen(i)[0] for i in range(n)]
        sig = o.g2_mul(hm, sum(sks) % o.R)
        assert o.fast_aggregate_verify(pks, msg, sig)
        bad = o.g2_mul(hm, (sum(sks) + 1) % o.R)
        assert not o.fast_aggregate_verify(pks, msg, bad)
        out.append({"n": n, "msg": hx(msg), "pks": hx(b"".join(o.g1_to_blst_affine(p) for p in pks)),
                    "sig": hx(o.g2_to_blst_affine(sig)), "bad_sig": hx(o.g2_to_blst_affine(bad))})
        print("fav", n, flush=True)
    return out


def main():
    which = sys.argv[1:] or ["fields", "h2c", "pairing", "batch", "msm", "fav"]
    gens = {"fields": gen_fields, "h2c": gen_h2c, "pairing": gen_pairing, "batch": gen_batch,
            "msm": gen_msm, "fav": gen_fav}
    for w in which:
        data = gens[w]()
        with open(os.path.join(HERE, w + ".json"), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print("wrote", w, flush=True)


if __name__ == "__main__":
    main()
