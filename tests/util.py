"""Shared helpers for tests: fixture loading and canonicalisation of projective outputs."""
import ctypes
import json
import os

import bls12381_py as o

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def buf(n):
    return ctypes.create_string_buffer(n)


def fp_int(b):
    return o.fp_from_mont_bytes(bytes(b))


def fp2_int(b):
    return (fp_int(b[:48]), fp_int(b[48:96]))


def g1_jac_to_affine(b):
    """144-byte Jacobian (Montgomery) -> oracle affine point."""
    x, y, z = fp_int(b[:48]), fp_int(b[48:96]), fp_int(b[96:144])
    if z == 0:
        return None
    zi = o.fp_inv(z)
    return (x * zi * zi % o.P, y * zi * zi * zi % o.P)


def g2_jac_to_affine(b):
    x, y, z = fp2_int(b[:96]), fp2_int(b[96:192]), fp2_int(b[192:288])
    if z == (0, 0):
        return None
    zi = o.f2inv(z)
    zi2 = o.f2sqr(zi)
    return (o.f2mul(x, zi2), o.f2mul(y, o.f2mul(zi2, zi)))


def g1_aff_to_jac_bytes(p):
    if p is None:
        return bytes(144)
    return o.g1_to_blst_affine(p) + o.fp_to_mont_bytes(1)


def g2_aff_to_jac_bytes(p):
    if p is None:
        return bytes(288)
    return o.g2_to_blst_affine(p) + o.fp_to_mont_bytes(1) + bytes(48)


def fp12_from_bytes(b):
    """576-byte blst_fp12 image -> oracle flat tuple."""
    t = [fp2_int(b[96 * i:96 * i + 96]) for i in range(6)]
    return (t[0], t[3], t[1], t[4], t[2], t[5])


def fp12_hexlist_to_flat(h):
    t = [(fp_int(bytes.fromhex(c[0])), fp_int(bytes.fromhex(c[1]))) for c in h]
    return (t[0], t[3], t[1], t[4], t[2], t[5])


def fp12_to_bytes(a):
    t = o.f12_to_tower_ints(a)
    return b"".join(o.fp_to_mont_bytes(c[0]) + o.fp_to_mont_bytes(c[1]) for c in t)
