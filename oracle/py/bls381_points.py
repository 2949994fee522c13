"""CPU oracle for the compressed point wire format and for point aggregation.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library imports this module.

The reference takes ark-ec points; its callers obtain them from bytes with ark-bls12-381 0.4.0
(/root/reference/Cargo.toml:13) `G1Affine` / `G2Affine` `deserialize_compressed` and write them with
`serialize_compressed`.  That crate is not vendored in the reference, so this module restates the
published encoding it implements (the ZCash BLS12-381 serialization, also IETF
draft-irtf-cfrg-pairing-friendly-curves appendix C):

  G1: 48 bytes, big-endian x.   G2: 96 bytes, big-endian x.c1 then x.c0.
  Top three bits of byte 0: 0x80 compressed form, 0x40 point at infinity (all other bits zero),
  0x20 y is the lexicographically larger of the two roots (Fq: y > (p-1)/2; Fq2: decided by c1, by
  c0 when c1 = 0).
  Decoding checks: x < p, x^3 + b is a square, and (optionally) membership in the order-r subgroup.

PARITY STATUS: "parity unpinned" against ark-bls12-381; pinned by the generators' well-known
encodings, by round trips, and by definition-level checks ([r]P = O with affine arithmetic, against
the endomorphism tests the device uses).
"""
from bls381_ref import (P, R_ORDER, G1_GEN, G2_GEN, FQ2_ZERO, fq_sqrt, fq2_sqrt, fq2_add, fq2_mul, fq2_square,
                        fq2_neg, g1_add, g1_mul, g2_add, g2_mul)

OK, INFINITY, BAD_ENCODING, NOT_ON_CURVE, NOT_IN_SUBGROUP = 0, 1, 2, 3, 4
HALF = (P - 1) // 2


def fq_is_larger(y):
    return y % P > HALF


def fq2_is_larger(y):
    return fq_is_larger(y[1]) if y[1] % P else fq_is_larger(y[0])


def g1_in_subgroup(p):
    return p is None or g1_mul(p, R_ORDER) is None


def g2_in_subgroup(p):
    return p is None or g2_mul(p, R_ORDER) is None


def g1_compress(p):
    if p is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray((p[0] % P).to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if fq_is_larger(p[1]) else 0)
    return bytes(b)


def g2_compress(p):
    if p is None:
        return bytes([0xC0]) + bytes(95)
    b = bytearray((p[0][1] % P).to_bytes(48, "big") + (p[0][0] % P).to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if fq2_is_larger(p[1]) else 0)
    return bytes(b)


def _flags(b):
    return b[0] >> 7, (b[0] >> 6) & 1, (b[0] >> 5) & 1


def g1_decompress(b, check_subgroup=True):
    """-> (status, point or None)."""
    assert len(b) == 48
    comp, inf, sort = _flags(b)
    x = int.from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:]), "big")
    if not comp:
        return BAD_ENCODING, None
    if inf:
        return (INFINITY, None) if (sort == 0 and x == 0) else (BAD_ENCODING, None)
    if x >= P:
        return BAD_ENCODING, None
    y = fq_sqrt((x * x * x + 4) % P)
    if y is None:
        return NOT_ON_CURVE, None
    if fq_is_larger(y) != bool(sort):
        y = (-y) % P
    pt = (x, y)
    if check_subgroup and not g1_in_subgroup(pt):
        return NOT_IN_SUBGROUP, None
    return OK, pt


def g2_decompress(b, check_subgroup=True):
    assert len(b) == 96
    comp, inf, sort = _flags(b)
    x1 = int.from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]), "big")
    x0 = int.from_bytes(bytes(b[48:]), "big")
    if not comp:
        return BAD_ENCODING, None
    if inf:
        return (INFINITY, None) if (sort == 0 and x0 == 0 and x1 == 0) else (BAD_ENCODING, None)
    if x0 >= P or x1 >= P:
        return BAD_ENCODING, None
    x = (x0, x1)
    y = fq2_sqrt(fq2_add(fq2_mul(fq2_square(x), x), (4, 4)))
    if y is None or y[1] is None or y[0] is None or fq2_square(y) != fq2_add(fq2_mul(fq2_square(x), x), (4, 4)):
        return NOT_ON_CURVE, None
    if fq2_is_larger(y) != bool(sort):
        y = fq2_neg(y)
    pt = (x, y)
    if check_subgroup and not g2_in_subgroup(pt):
        return NOT_IN_SUBGROUP, None
    return OK, pt


def g1_aggregate(points):
    acc = None
    for p in points:
        acc = g1_add(acc, p)
    return acc


def g2_aggregate(points):
    acc = None
    for p in points:
        acc = g2_add(acc, p)
    return acc


# Well-known encodings of the standard generators (ZCash bls12_381 crate documentation / IETF draft).
G1_GEN_COMPRESSED = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G2_GEN_COMPRESSED = bytes.fromhex(
    "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
