"""Big-integer restatement of the BLS12-381 batch-verification path of nim-blscurve.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or executed
by the product (``nim-blscurve_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it, and only as the checker.

What it restates (reference = /root/reference, nim-blscurve @ 2025-10-31):
  * the call sequence / blinding-scalar derivation / chunking of
    ``blscurve/bls_batch_verifier.nim:121-160,296-371``,
    ``blscurve/blst/blst_min_pubkey_sig_core.nim:476-568,570-647,649-672``,
    ``blscurve/parallel_chunks.nim:42-66``, ``blscurve/blst/sha256_abi.nim:52-74``;
  * the arithmetic the reference delegates to supranational/blst (git submodule
    ``vendor/blst``, NOT checked out in the snapshot, pinned SHA unrecoverable,
    comment at ``blst_min_pubkey_sig_core.nim:609`` links v0.3.13).  That arithmetic is
    restated from the published specifications: RFC 9380 (hash_to_curve suite
    BLS12381G2_XMD:SHA-256_SSWU_RO_), draft-irtf-cfrg-bls-signature-05 (PoP scheme,
    KeyGen), the ZCash serialisation format and the optimal-ate pairing on BLS12-381
    (parameters at ``blst_min_pubkey_sig_core.nim:362-367``,
    ``tests/priv_to_pub.sage:15-50``).

Parity pins (all checked by ``tests/test_oracle_kats.py``):
  * sk -> compressed pk, 6 KATs        ``tests/priv_to_pub.nim:32-81``
  * sk -> affine (x, y), 10 KATs       ``tests/priv_to_pub.sage:76-124``
  * ikm -> sk (KeyGen), 1 KAT          ``tests/priv_to_pub.nim:57-81``
  * (sk, pk, PoP proof) x3             ``tests/eth2_vectors.nim:33-69``  -> pins
    expand_message_xmd + hash_to_field + SSWU + 3-isogeny + cofactor clearing + G2
    scalar mult + G2 compression byte-for-byte, and popVerify true / cross-key false
    pins Miller loop + final exponentiation verdicts.
  * zero signature == c0 00..00, malformed 96-byte signature rejected
                                       ``tests/serialization.nim:19-45``
  * every boolean scenario of          ``tests/t_batch_verifier.nim:65-274``
The raw Miller-loop value and r_i-dependent intermediates are not pinned by the
reference; only canonical (affine/compressed) group elements,
final-exponentiated GT values and verdicts are parity targets.
"""
import hashlib
import hmac

# ----------------------------------------------------------------------------
# Parameters (tests/priv_to_pub.sage:15-50; blst_min_pubkey_sig_core.nim:362-367)
# ----------------------------------------------------------------------------
X_ABS = 0xd201000000010000            # |x|, x is negative
X = -X_ABS
P = (X - 1) ** 2 * (X ** 4 - X ** 2 + 1) // 3 + X
R = X ** 4 - X ** 2 + 1
assert P == 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
assert R == 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
H1 = 0x396c8c005555e1568c00aaab0000aaab
H_EFF_G2 = 0xbc69f08f2ee75b3584c6a0ea91b352888e2a8e9145ad7689986ff031508ffe1329c2f178731db956d82bf015d1212b02ec0ec69d7477c1ae954cbc06689f6a359894c0adebbf6b4e8020005aaa95551

G1_GEN = (
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
)
G2_GEN = (
    (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
     0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
    (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
     0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
)

DST_SIG = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"   # bls_sig_min_pubkey.nim:31
DST_POP = b"BLS_POP_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"   # bls_sig_min_pubkey.nim:32

MONT_R = (1 << 384) % P            # blst in-memory form: a*2^384 mod p, 6 LE u64 limbs


def sha256(b):
    return hashlib.sha256(b).digest()


# ----------------------------------------------------------------------------
# Fp, Fp2 = Fp[u]/(u^2+1)
# ----------------------------------------------------------------------------
def fp_inv(a):
    return pow(a, P - 2, P)


def fp_sqrt(a):
    """p = 3 mod 4."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)            # non-residue 1+u (tests/priv_to_pub.sage:33)


def f2(a0, a1=0):
    return (a0 % P, a1 % P)


def f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2neg(a):
    return (-a[0] % P, -a[1] % P)


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2muls(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2conj(a):
    return (a[0], -a[1] % P)


def f2inv(a):
    n = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * n % P, -a[1] * n % P)


def f2pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2sqr(a)
        e >>= 1
    return r


def f2_is_zero(a):
    return a[0] % P == 0 and a[1] % P == 0


def f2_is_square(a):
    n = (a[0] * a[0] + a[1] * a[1]) % P          # norm to Fp
    return n == 0 or pow(n, (P - 1) // 2, P) == 1


def f2sqrt(a):
    """Any square root in Fp2 or None (complex method, p = 3 mod 4)."""
    if f2_is_zero(a):
        return F2_ZERO
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt(-a0 % P)
        return (0, s)
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    half = fp_inv(2)
    d = (a0 + n) * half % P
    x0 = fp_sqrt(d)
    if x0 is None:
        d = (a0 - n) * half % P
        x0 = fp_sqrt(d)
        if x0 is None:
            return None
    x1 = a1 * fp_inv(2 * x0 % P) % P
    r = (x0, x1)
    return r if f2sqr(r) == (a0 % P, a1 % P) else None


def f2sgn0(a):
    """RFC 9380 section 4.1, m = 2."""
    s0 = a[0] & 1
    z0 = a[0] == 0
    s1 = a[1] & 1
    return s0 | (z0 & s1)


# ----------------------------------------------------------------------------
# Fp12 = Fp2[w]/(w^6 - xi), flat basis w^0..w^5.
# Tower view used by the product (Fp6 = Fp2[v]/(v^3-xi), Fp12 = Fp6[w]/(w^2-v)):
#   c0 = (a0, a1, a2), c1 = (b0, b1, b2)  <->  flat (a0, b0, a1, b1, a2, b2)
# ----------------------------------------------------------------------------
F12_ONE = (F2_ONE,) + (F2_ZERO,) * 5


def f12mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        ai = a[i]
        if ai == F2_ZERO:
            continue
        for j in range(6):
            bj = b[j]
            if bj == F2_ZERO:
                continue
            t[i + j] = f2add(t[i + j], f2mul(ai, bj))
    for k in range(10, 5, -1):
        t[k - 6] = f2add(t[k - 6], f2mul(t[k], XI))
    return tuple(t[:6])


def f12sqr(a):
    return f12mul(a, a)


def f12conj(a):
    """a^(p^6): w -> -w."""
    return (a[0], f2neg(a[1]), a[2], f2neg(a[3]), a[4], f2neg(a[5]))


_FROB_GAMMA = [f2pow(XI, i * (P - 1) // 6) for i in range(6)]


def f12frob(a):
    """a^p."""
    return tuple(f2mul(f2conj(a[i]), _FROB_GAMMA[i]) for i in range(6))


def f12frob_n(a, n):
    for _ in range(n):
        a = f12frob(a)
    return a


def f12pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12mul(r, a)
        a = f12mul(a, a)
        e >>= 1
    return r


# Fp6 helpers on (a0,a1,a2), v^3 = xi  (needed only for inversion)
def _f6mul(a, b):
    t = [F2_ZERO] * 5
    for i in range(3):
        for j in range(3):
            t[i + j] = f2add(t[i + j], f2mul(a[i], b[j]))
    return (f2add(t[0], f2mul(t[3], XI)), f2add(t[1], f2mul(t[4], XI)), t[2])


def _f6mul_by_v(a):
    return (f2mul(a[2], XI), a[0], a[1])


def _f6inv(a):
    a0, a1, a2 = a
    c0 = f2sub(f2sqr(a0), f2mul(XI, f2mul(a1, a2)))
    c1 = f2sub(f2mul(XI, f2sqr(a2)), f2mul(a0, a1))
    c2 = f2sub(f2sqr(a1), f2mul(a0, a2))
    t = f2add(f2mul(a0, c0), f2mul(XI, f2add(f2mul(a2, c1), f2mul(a1, c2))))
    ti = f2inv(t)
    return (f2mul(c0, ti), f2mul(c1, ti), f2mul(c2, ti))


def f12inv(a):
    c0 = (a[0], a[2], a[4])
    c1 = (a[1], a[3], a[5])
    d = tuple(f2sub(x, y) for x, y in zip(_f6mul(c0, c0), _f6mul_by_v(_f6mul(c1, c1))))
    di = _f6inv(d)
    r0 = _f6mul(c0, di)
    r1 = tuple(f2neg(x) for x in _f6mul(c1, di))
    return (r0[0], r1[0], r0[1], r1[1], r0[2], r1[2])


# ----------------------------------------------------------------------------
# Curves.  Points are None (infinity) or affine tuples.
#   E1/Fp : y^2 = x^3 + 4          E2/Fp2 : y^2 = x^3 + 4(1+u)
# Generic affine group law parameterised by field ops.
# ----------------------------------------------------------------------------
class _Field:
    pass


FP = _Field()
FP.add = lambda a, b: (a + b) % P
FP.sub = lambda a, b: (a - b) % P
FP.mul = lambda a, b: a * b % P
FP.sqr = lambda a: a * a % P
FP.inv = fp_inv
FP.neg = lambda a: -a % P
FP.zero = 0
FP.smul = lambda a, k: a * k % P

FP2 = _Field()
FP2.add = f2add
FP2.sub = f2sub
FP2.mul = f2mul
FP2.sqr = f2sqr
FP2.inv = f2inv
FP2.neg = f2neg
FP2.zero = F2_ZERO
FP2.smul = f2muls

B1 = 4
B2 = (4, 4)


def ec_add(F, p, q, a=None):
    """Affine addition on y^2 = x^3 + a x + b (b not needed)."""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if F.add(y1, y2) == F.zero:
            return None
        num = F.smul(F.sqr(x1), 3)
        if a is not None:
            num = F.add(num, a)
        lam = F.mul(num, F.inv(F.smul(y1, 2)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, p):
    return None if p is None else (p[0], F.neg(p[1]))


def ec_mul(F, p, k, a=None):
    if k < 0:
        return ec_mul(F, ec_neg(F, p), -k, a)
    r = None
    while k:
        if k & 1:
            r = ec_add(F, r, p, a)
        p = ec_add(F, p, p, a)
        k >>= 1
    return r


def g1_add(p, q):
    return ec_add(FP, p, q)


def g1_mul(p, k):
    return ec_mul(FP, p, k)


def g1_neg(p):
    return ec_neg(FP, p)


def g2_add(p, q):
    return ec_add(FP2, p, q)


def g2_mul(p, k):
    return ec_mul(FP2, p, k)


def g2_neg(p):
    return ec_neg(FP2, p)


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - B1) % P == 0


def g2_on_curve(p):
    if p is None:
        return True
    x, y = p
    return f2sub(f2sqr(y), f2add(f2mul(f2sqr(x), x), B2)) == F2_ZERO


def g1_in_subgroup(p):
    return g1_on_curve(p) and g1_mul(p, R) is None


def g2_in_subgroup(p):
    return g2_on_curve(p) and g2_mul(p, R) is None


# psi endomorphism on E2 (untwist-Frobenius-twist), used by the fast cofactor clearing
_PSI_CX = f2inv(f2pow(XI, (P - 1) // 3))
_PSI_CY = f2inv(f2pow(XI, (P - 1) // 2))


def g2_psi(p):
    if p is None:
        return None
    return (f2mul(f2conj(p[0]), _PSI_CX), f2mul(f2conj(p[1]), _PSI_CY))


# ----------------------------------------------------------------------------
# Serialisation (ZCash format; tests/priv_to_pub.sage:59-65, tests/serialization.nim:19-45)
# ----------------------------------------------------------------------------
def g1_compress(p):
    if p is None:
        return bytes([0xc0]) + bytes(47)
    x, y = p
    v = x | (1 << 383)
    if y > P - y:
        v |= 1 << 381
    return v.to_bytes(48, "big")


def g1_decompress(b):
    """-> point, or raises ValueError (bls_sig_io.nim:81-99 semantics minus subgroup check)."""
    if len(b) != 48 or not b[0] & 0x80:
        raise ValueError("bad encoding")
    if b[0] & 0x40:
        if any(b[1:]) or b[0] & 0x3f:
            raise ValueError("bad infinity")
        return None
    sign = (b[0] >> 5) & 1
    x = int.from_bytes(b, "big") & ((1 << 381) - 1)
    if x >= P:
        raise ValueError("x >= p")
    y = fp_sqrt((x ** 3 + B1) % P)
    if y is None:
        raise ValueError("not on curve")
    if (y > P - y) != bool(sign):
        y = P - y
    return (x, y)


def _f2_lex_largest(y):
    if y[1] != 0:
        return y[1] > P - y[1]
    return y[0] > P - y[0]


def g2_compress(p):
    if p is None:
        return bytes([0xc0]) + bytes(95)
    x, y = p
    v = x[1] | (1 << 383)
    if _f2_lex_largest(y):
        v |= 1 << 381
    return v.to_bytes(48, "big") + x[0].to_bytes(48, "big")


def g2_decompress(b):
    if len(b) != 96 or not b[0] & 0x80:
        raise ValueError("bad encoding")
    if b[0] & 0x40:
        if any(b[1:]) or b[0] & 0x3f:
            raise ValueError("bad infinity")
        return None
    sign = (b[0] >> 5) & 1
    x1 = int.from_bytes(b[:48], "big") & ((1 << 381) - 1)
    x0 = int.from_bytes(b[48:], "big")
    if x0 >= P or x1 >= P:
        raise ValueError("x >= p")
    x = (x0, x1)
    y = f2sqrt(f2add(f2mul(f2sqr(x), x), B2))
    if y is None:
        raise ValueError("not on curve")
    if _f2_lex_largest(y) != bool(sign):
        y = f2neg(y)
    return (x, y)


def g1_serialize(p):
    """Uncompressed 96-byte form (blst_p1_affine_serialize; `serialize` into array[96, byte], bls_sig_io.nim:194-245)."""
    if p is None:
        return bytes([0x40]) + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def g1_deserialize(b):
    """blst_p1_deserialize semantics (bls_sig_io.nim:88-91 for 96-byte input): top bits 000 uncompressed, 1xx compressed
    (first 48 bytes), 01x infinity (0x40 then zeros only)."""
    if len(b) != 96:
        raise ValueError("length")
    if b[0] & 0x80:
        return g1_decompress(b[:48])
    if b[0] & 0x40:
        if b[0] & 0x3f or any(b[1:]):
            raise ValueError("bad infinity")
        return None
    if b[0] & 0x20:
        raise ValueError("bad encoding")
    x, y = int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big")
    if x >= P or y >= P:
        raise ValueError("coordinate >= p")
    if not g1_on_curve((x, y)):
        raise ValueError("not on curve")
    return (x, y)


def g2_serialize(p):
    if p is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = p
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


def g2_deserialize(b):
    """blst_p2_deserialize semantics (bls_sig_io.nim:49-52 for 192-byte input)."""
    if len(b) != 192:
        raise ValueError("length")
    if b[0] & 0x80:
        return g2_decompress(b[:96])
    if b[0] & 0x40:
        if b[0] & 0x3f or any(b[1:]):
            raise ValueError("bad infinity")
        return None
    if b[0] & 0x20:
        raise ValueError("bad encoding")
    x1, x0, y1, y0 = (int.from_bytes(b[48 * i:48 * i + 48], "big") for i in range(4))
    if max(x0, x1, y0, y1) >= P:
        raise ValueError("coordinate >= p")
    q = ((x0, x1), (y0, y1))
    if not g2_on_curve(q):
        raise ValueError("not on curve")
    return q


# blst in-memory layouts (blst_abi.nim:87-122): Montgomery limbs, little-endian
def fp_to_mont_bytes(a):
    return (a * MONT_R % P).to_bytes(48, "little")


def fp_from_mont_bytes(b):
    return int.from_bytes(b, "little") * fp_inv(MONT_R) % P


def g1_to_blst_affine(p):
    """96 bytes; infinity = all zero (blst_lowlevel.nim:28-44)."""
    if p is None:
        return bytes(96)
    return fp_to_mont_bytes(p[0]) + fp_to_mont_bytes(p[1])


def g1_from_blst_affine(b):
    if not any(b):
        return None
    return (fp_from_mont_bytes(b[:48]), fp_from_mont_bytes(b[48:96]))


def g2_to_blst_affine(p):
    """192 bytes: x.c0, x.c1, y.c0, y.c1 (blst_abi.nim:96-98 'real, imaginary')."""
    if p is None:
        return bytes(192)
    (x0, x1), (y0, y1) = p
    return b"".join(fp_to_mont_bytes(v) for v in (x0, x1, y0, y1))


def g2_from_blst_affine(b):
    if not any(b):
        return None
    v = [fp_from_mont_bytes(b[48 * i:48 * i + 48]) for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def signature_set_bytes(pk, msg32, sig):
    """The reference's SignatureSet tuple, 320 B (pk@0, msg@96, sig@128)."""
    assert len(msg32) == 32
    return g1_to_blst_affine(pk) + msg32 + g2_to_blst_affine(sig)


# ----------------------------------------------------------------------------
# hash_to_curve, suite BLS12381G2_XMD:SHA-256_SSWU_RO_  (RFC 9380 sections 5, 6.6.3, 8.8.2)
# ----------------------------------------------------------------------------
def expand_message_xmd(msg, dst, n):
    ell = (n + 31) // 32
    assert ell <= 255 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime)
    bi = sha256(b0 + b"\x01" + dst_prime)
    out = bi
    for i in range(2, ell + 1):
        bi = sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime)
        out += bi
    return out[:n]


def hash_to_field_fp2(msg, dst, count=2):
    L = 64
    u = expand_message_xmd(msg, dst, count * 2 * L)
    out = []
    for i in range(count):
        e = []
        for j in range(2):
            off = L * (j + i * 2)
            e.append(int.from_bytes(u[off:off + L], "big") % P)
        out.append(tuple(e))
    return out


SSWU_A = (0, 240)
SSWU_B = (1012, 1012)
SSWU_Z = (P - 2, P - 1)          # -(2 + u)


def sswu_g2(u):
    """Simplified SWU onto E2': y^2 = x^3 + A'x + B' (RFC 9380 section 6.6.2)."""
    A, B, Z = SSWU_A, SSWU_B, SSWU_Z
    zu2 = f2mul(Z, f2sqr(u))
    tv1 = f2add(f2sqr(zu2), zu2)
    if f2_is_zero(tv1):
        x1 = f2mul(B, f2inv(f2mul(Z, A)))
    else:
        x1 = f2mul(f2mul(f2neg(B), f2inv(A)), f2add(F2_ONE, f2inv(tv1)))
    gx1 = f2add(f2add(f2mul(f2sqr(x1), x1), f2mul(A, x1)), B)
    if f2_is_square(gx1):
        x, y = x1, f2sqrt(gx1)
    else:
        x2 = f2mul(zu2, x1)
        gx2 = f2add(f2add(f2mul(f2sqr(x2), x2), f2mul(A, x2)), B)
        x, y = x2, f2sqrt(gx2)
    assert y is not None
    if f2sgn0(u) != f2sgn0(y):
        y = f2neg(y)
    return (x, y)


def _derive_iso3():
    """3-isogeny E2' -> E2 (RFC 9380 appendix E.3), re-derived with Velu's formulas.

    The kernel is {O, +-K} with x(K) = -6 + 6u (the double root of the RFC's x_den
    = x^2 + (12 - 12u) x - 72u).  Velu gives a normalised isogeny onto
    y^2 = x^3 + b'' ; the isomorphism (x, y) -> (x/s^2, y/s^3) with s^6 = b''/(4(1+u))
    lands on E2.  j(E2) = 0, so six choices of s exist; the RFC's is the one with
    1/s^2 = k_(1,3) and 1/s^3 = k_(3,3) below, and the three PoP KATs of
    tests/eth2_vectors.nim:33-47 confirm the choice end to end.
    """
    A, B = SSWU_A, SSWU_B
    xk = (P - 6, 6)
    yk2 = f2add(f2add(f2mul(f2sqr(xk), xk), f2mul(A, xk)), B)      # y_K^2
    gx = f2add(f2muls(f2sqr(xk), 3), A)
    vq = f2muls(gx, 2)
    uq = f2muls(yk2, 4)
    v = vq
    w = f2add(uq, f2mul(xk, vq))
    a_img = f2sub(A, f2muls(v, 5))
    b_img = f2sub(B, f2muls(w, 7))
    assert a_img == F2_ZERO
    # X(x) = x + vq/(x-xk) + uq/(x-xk)^2 ; Y = y * dX/dx
    k13 = 0x171d6541fa38ccfaed6dea691f5fb614cb14b4e7f4e810aa22d6108f142b85757098e38d0f671c7188e2aaaaaaaa5ed1
    k33 = 0x124c9ad43b6cf79bfbf7043de3811ad0761b0f37a1e26286b0e977c69aa274524e79097a56dc4bd9e1b371c71c718b10
    inv_s2 = (k13, 0)
    inv_s3 = (k33, 0)
    # consistency: (1/s^2)^3 == (1/s^3)^2 and b_img * (1/s^3)^2 == 4(1+u)
    assert f2mul(f2sqr(inv_s2), inv_s2) == f2sqr(inv_s3)
    assert f2mul(b_img, f2sqr(inv_s3)) == B2
    return xk, vq, uq, inv_s2, inv_s3


_ISO_XK, _ISO_VQ, _ISO_UQ, _ISO_INV_S2, _ISO_INV_S3 = _derive_iso3()


def iso3_g2(p):
    """E2' -> E2."""
    if p is None:
        return None
    x, y = p
    d = f2sub(x, _ISO_XK)
    if f2_is_zero(d):
        return None
    di = f2inv(d)
    di2 = f2sqr(di)
    di3 = f2mul(di2, di)
    X = f2add(x, f2add(f2mul(_ISO_VQ, di), f2mul(_ISO_UQ, di2)))
    dX = f2sub(F2_ONE, f2add(f2mul(_ISO_VQ, di2), f2muls(f2mul(_ISO_UQ, di3), 2)))
    Y = f2mul(y, dX)
    return (f2mul(X, _ISO_INV_S2), f2mul(Y, _ISO_INV_S3))


def iso3_coefficients():
    """The RFC 9380 E.3 polynomial coefficients (k_(1,i), k_(2,i), k_(3,i), k_(4,i)), low
    degree first, recovered from the Velu form.  Used to hand constants to the product."""
    xk, vq, uq = _ISO_XK, _ISO_VQ, _ISO_UQ
    nx = f2neg(xk)
    # (x - xk)^2 = x^2 + 2nx x + nx^2 ; (x-xk)^3
    d2 = [f2sqr(nx), f2muls(nx, 2), F2_ONE]
    d3 = [f2mul(f2sqr(nx), nx), f2muls(f2sqr(nx), 3), f2muls(nx, 3), F2_ONE]
    # x_num = x*(x-xk)^2 + vq (x-xk) + uq
    xn = [F2_ZERO] + d2
    xn[0] = f2add(xn[0], f2add(f2mul(vq, nx), uq))
    xn[1] = f2add(xn[1], vq)
    xn = [f2mul(c, _ISO_INV_S2) for c in xn]
    # y_num = (x-xk)^3 - vq (x-xk) - 2 uq
    yn = list(d3)
    yn[0] = f2sub(yn[0], f2add(f2mul(vq, nx), f2muls(uq, 2)))
    yn[1] = f2sub(yn[1], vq)
    yn = [f2mul(c, _ISO_INV_S3) for c in yn]
    return xn, d2, yn, d3


def clear_cofactor_g2_slow(p):
    return g2_mul(p, H_EFF_G2)


def clear_cofactor_g2(p):
    """RFC 9380 appendix G.3 (Budroni-Pintore): [x^2-x-1]P + [x-1]psi(P) + psi^2(2P)."""
    t1 = g2_mul(p, X)                    # x P
    t2 = g2_psi(p)
    t3 = g2_psi(g2_psi(g2_add(p, p)))    # psi^2(2P)
    t3 = g2_add(t3, g2_neg(t2))          # psi^2(2P) - psi(P)
    t2 = g2_add(t1, t2)                  # xP + psi(P)
    t2 = g2_mul(t2, X)                   # x^2 P + x psi(P)
    t3 = g2_add(t3, t2)
    t3 = g2_add(t3, g2_neg(t1))
    return g2_add(t3, g2_neg(p))


def hash_to_g2(msg, dst=DST_SIG):
    """blst_hash_to_g2(msg, dst, aug='') (blst_abi.nim:383)."""
    u0, u1 = hash_to_field_fp2(msg, dst, 2)
    q0 = sswu_g2(u0)
    q1 = sswu_g2(u1)
    rr = ec_add(FP2, q0, q1, SSWU_A)     # add on E2' (homomorphism: same as adding after iso)
    return clear_cofactor_g2(iso3_g2(rr))


# ----------------------------------------------------------------------------
# Pairing: optimal ate, Miller loop over |x| with affine twist arithmetic,
# line l*w^3 = (lam*xt - yt) - lam*xp*w^2 + yp*w^3  (untwist (x,y)->(x/w^2, y/w^3)).
# ----------------------------------------------------------------------------
def _line(lam, t, p):
    xt, yt = t
    xp, yp = p
    return (f2sub(f2mul(lam, xt), yt), F2_ZERO, f2muls(f2neg(lam), xp), (yp, 0), F2_ZERO, F2_ZERO)


def miller_loop(pairs):
    """prod_i f_{|x|,Q_i}(P_i), conjugated (x<0).  pairs: [(P in G1, Q in G2)]; infinity pairs skipped."""
    pairs = [(p, q) for p, q in pairs if p is not None and q is not None]
    f = F12_ONE
    ts = [q for _, q in pairs]
    for bit in range(X_ABS.bit_length() - 2, -1, -1):
        f = f12sqr(f)
        for i, (p, q) in enumerate(pairs):
            t = ts[i]
            lam = f2mul(f2muls(f2sqr(t[0]), 3), f2inv(f2muls(t[1], 2)))
            f = f12mul(f, _line(lam, t, p))
            ts[i] = g2_add(t, t)
        if (X_ABS >> bit) & 1:
            for i, (p, q) in enumerate(pairs):
                t = ts[i]
                lam = f2mul(f2sub(q[1], t[1]), f2inv(f2sub(q[0], t[0])))
                f = f12mul(f, _line(lam, t, p))
                ts[i] = g2_add(t, q)
    return f12conj(f)


FINAL_EXP = (P ** 12 - 1) // R
HARD_EXP = (P ** 4 - P ** 2 + 1) // R


def final_exp_naive(f):
    return f12pow(f, FINAL_EXP)


def _cyc_exp_x(a):
    """a^x for a in the cyclotomic subgroup (x<0: conjugate = inverse)."""
    return f12conj(f12pow(a, X_ABS))


def final_exp(f):
    """f^((p^12-1)/r * 3)   -- easy part then Hayashida-Hayasaka-Teruya hard part:
    3*(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3."""
    t = f12mul(f12conj(f), f12inv(f))              # f^(p^6-1)
    t = f12mul(f12frob_n(t, 2), t)                 # ^(p^2+1)
    a = f12mul(_cyc_exp_x(t), f12conj(t))          # t^(x-1)
    a = f12mul(_cyc_exp_x(a), f12conj(a))          # t^((x-1)^2)
    b = f12mul(_cyc_exp_x(a), f12frob(a))          # ^(x+p)
    c = f12mul(f12mul(_cyc_exp_x(_cyc_exp_x(b)), f12frob_n(b, 2)), f12conj(b))   # ^(x^2+p^2-1)
    return f12mul(c, f12mul(f12sqr(t), t))


def pairing(p, q):
    return final_exp(miller_loop([(p, q)]))


# ----------------------------------------------------------------------------
# Scheme layer (bls_sig_min_pubkey.nim; draft-irtf-cfrg-bls-signature) -- only to make inputs
# ----------------------------------------------------------------------------
def keygen(ikm, info=b""):
    """blst_keygen (bls_spec_keygen_blst.nim:73).  The KAT at tests/priv_to_pub.nim:57-81 pins the
    draft-irtf-cfrg-bls-signature-02/03 flavour: salt "BLS-SIG-KEYGEN-SALT-" used as is (not hashed),
    PRK = HKDF-Extract(salt, IKM || 0x00), OKM = HKDF-Expand(PRK, info || I2OSP(48, 2), 48)."""
    assert len(ikm) >= 32
    prk = hmac.new(b"BLS-SIG-KEYGEN-SALT-", ikm + b"\x00", hashlib.sha256).digest()
    L = 48
    okm = b""
    t = b""
    i = 1
    while len(okm) < L:
        t = hmac.new(prk, t + info + L.to_bytes(2, "big") + bytes([i]), hashlib.sha256).digest()
        okm += t
        i += 1
    return int.from_bytes(okm[:L], "big") % R


def keygen_seed(seed):
    """tests/t_batch_verifier.nim:34-38: ikm[0..8] = LE64(seed), rest zero."""
    sk = keygen(seed.to_bytes(8, "little") + bytes(24))
    return g1_mul(G1_GEN, sk), sk


def sk_to_pk(sk):
    return g1_mul(G1_GEN, sk)


def sign(sk, msg, dst=DST_SIG):
    """coreSign (blst_min_pubkey_sig_core.nim:230-251)."""
    return g2_mul(hash_to_g2(msg, dst), sk)


def pop_prove(sk):
    return sign(sk, g1_compress(sk_to_pk(sk)), DST_POP)


def core_verify(pk, msg, sig, dst=DST_SIG):
    """coreVerifyNoGroupCheck (blst_min_pubkey_sig_core.nim:269-297): e(pk,H(m)) == e(G1,sig)."""
    if pk is None:
        return False
    f = miller_loop([(pk, hash_to_g2(msg, dst)), (g1_neg(G1_GEN), sig)])
    return final_exp(f) == F12_ONE


def pop_verify(pk, proof):
    return core_verify(pk, g1_compress(pk), proof, DST_POP)


def aggregate_g1(pks):
    acc = None
    for p in pks:
        acc = g1_add(acc, p)
    return acc


def aggregate_g2(sigs):
    acc = None
    for s in sigs:
        acc = g2_add(acc, s)
    return acc


def fast_aggregate_verify(pks, msg, sig):
    """bls_sig_min_pubkey.nim:234-258."""
    if len(pks) == 0:
        return False
    return core_verify(aggregate_g1(pks), msg, sig)


def aggregate_verify(pks, msgs, sig):
    """bls_sig_min_pubkey.nim:127-199 shape (used only by the forged-pair construction check)."""
    if len(pks) == 0 or len(pks) != len(msgs):
        return False
    pairs = [(pk, hash_to_g2(m)) for pk, m in zip(pks, msgs)]
    pairs.append((g1_neg(G1_GEN), sig))
    return final_exp(miller_loop(pairs)) == F12_ONE


# ----------------------------------------------------------------------------
# Batch verification (THE path)
# ----------------------------------------------------------------------------
def parallel_chunks(num_chunks, total):
    """parallel_chunks.nim:42-66 -> [(offset, len)] for chunk ids 0..num_chunks-1 (skips c >= total)."""
    base, rem = divmod(total, num_chunks)
    out = []
    for c in range(num_chunks):
        if c >= total:
            break
        if c < rem:
            out.append(((base + 1) * c, base + 1))
        else:
            out.append((base * c + rem, base))
    return out


def blinding_seed(rnd32, chunk_id=None):
    """ContextMultiAggregateVerify.init (core :476-505); tag = LE64(chunkID) (bls_batch_verifier.nim:335)."""
    if chunk_id is None:
        return sha256(rnd32)
    return sha256(rnd32 + chunk_id.to_bytes(8, "little"))


def blinding_next(seed):
    """update()'s scalar part (core :545-556): advance BEFORE use, skip zero low-u64."""
    while True:
        seed = sha256(seed)
        r = int.from_bytes(seed[:8], "little")
        if r != 0:
            return seed, r


def blinding_scalars(rnd32, n, num_chunks=None):
    """r_i for every tuple.  num_chunks=None -> serial path (batchVerifySerial), else the
    parallel path with B = min(n, num_chunks) contexts (bls_batch_verifier.nim:316)."""
    out = [0] * n
    if num_chunks is None:
        seed = blinding_seed(rnd32)
        for i in range(n):
            seed, out[i] = blinding_next(seed)
        return out
    b = min(n, num_chunks)
    for c, (off, ln) in enumerate(parallel_chunks(b, n)):
        seed = blinding_seed(rnd32, c)
        for j in range(ln):
            seed, out[off + j] = blinding_next(seed)
    return out


def batch_verify_stages(sets, rnd32, num_chunks=None):
    """Returns dict of canonical per-stage values + verdict for sets = [(pk, msg32, sig)].

    Semantics of blst_pairing_chk_n_mul_n_aggr_pk_in_g1 + commit + merge + finalverify as
    used by the reference:
        finalexp( prod_i ML(H(m_i), [r_i]PK_i) * ML(sum_i [r_i]S_i, -G1) ) == 1
    pk = infinity -> false (BLST_PK_IS_INFINITY); sig = infinity contributes nothing.
    """
    n = len(sets)
    if n == 0:
        return {"verdict": False}
    rs = blinding_scalars(rnd32, n, num_chunks)
    st = {"r": rs, "H": [], "rPK": [], "verdict": False}
    agg = None
    pairs = []
    for (pk, msg, sig), r in zip(sets, rs):
        if pk is None:
            return st
        agg = g2_add(agg, g2_mul(sig, r))
        h = hash_to_g2(msg)
        rpk = g1_mul(pk, r)
        st["H"].append(h)
        st["rPK"].append(rpk)
        pairs.append((rpk, h))
    st["aggsig"] = agg
    pairs.append((g1_neg(G1_GEN), agg))
    gt = final_exp(miller_loop(pairs))
    st["gt"] = gt
    st["verdict"] = gt == F12_ONE
    return st


def batch_verify(sets, rnd32, num_chunks=None):
    return batch_verify_stages(sets, rnd32, num_chunks)["verdict"]


def combine_scalars(rnd32, n):
    """combine()'s scalar derivation (core :588-606): words 3,2,1,0 of each digest, zeros skipped."""
    seed = rnd32
    avail = 0
    out = []
    for _ in range(n):
        while True:
            if avail == 0:
                seed = sha256(seed)
                avail = 4
            avail -= 1
            w = int.from_bytes(seed[8 * avail:8 * avail + 8], "little")
            if w != 0:
                out.append(w)
                break
    return out


def combine(rnd32, pks, sigs):
    """core :570-647."""
    assert len(pks) == len(sigs) and len(pks) > 0
    if len(pks) == 1:
        return pks[0], sigs[0]
    ss = combine_scalars(rnd32, len(pks))
    pk = None
    sg = None
    for s, p, q in zip(ss, pks, sigs):
        pk = g1_add(pk, g1_mul(p, s))
        sg = g2_add(sg, g2_mul(q, s))
    return pk, sg


def msm_g1(points, scalars, nbits=255):
    """blst_p1s_mult_pippenger semantics: sum [k_i mod 2^nbits] P_i (blst_abi.nim:336-340)."""
    acc = None
    mask = (1 << nbits) - 1
    for p, k in zip(points, scalars):
        acc = g1_add(acc, g1_mul(p, k & mask))
    return acc


def f12_to_tower_ints(a):
    """flat -> [c0.a0, c0.a1, c0.a2, c1.b0, c1.b1, c1.b2] each (re, im): blst_fp12 memory order."""
    return [a[0], a[2], a[4], a[1], a[3], a[5]]
