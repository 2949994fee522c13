"""Shared test helpers: oracle values <-> the ABI's limb arrays, seeded input generation."""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "oracle", "py"))
import bls381_ref as R  # noqa: E402


def fq_to_u64(a):
    """canonical int -> 6 u64 Montgomery limbs"""
    return R.fq_to_u64_limbs(R.to_mont(a))


def u64_to_fq(l):
    return R.from_mont(R.u64_limbs_to_int(l))


def g1_array(points):
    out = np.zeros((len(points), 12), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is not None:
            out[i] = fq_to_u64(p[0]) + fq_to_u64(p[1])
    return out


def g2_array(points):
    out = np.zeros((len(points), 24), dtype=np.uint64)
    for i, q in enumerate(points):
        if q is not None:
            out[i] = fq_to_u64(q[0][0]) + fq_to_u64(q[0][1]) + fq_to_u64(q[1][0]) + fq_to_u64(q[1][1])
    return out


def inf_flags(ps, qs):
    return np.array([(1 if p is None else 0) | (2 if q is None else 0) for p, q in zip(ps, qs)], dtype=np.uint8)


def fq12_array(vals):
    out = np.zeros((len(vals), 72), dtype=np.uint64)
    for i, v in enumerate(vals):
        flat = R.fq12_flatten_ark(v)
        out[i] = [x for c in flat for x in fq_to_u64(c)]
    return out


def fq12_from_row(row):
    row = [int(x) for x in row]
    return R.fq12_from_flat_ark([u64_to_fq(row[6 * k:6 * k + 6]) for k in range(12)])


def random_pairs(n, seed):
    rng = random.Random(seed)
    ps = [R.g1_mul(R.G1_GEN, rng.randrange(1, R.R_ORDER)) for _ in range(n)]
    qs = [R.g2_mul(R.G2_GEN, rng.randrange(1, R.R_ORDER)) for _ in range(n)]
    return ps, qs


def random_fq12(n, seed):
    rng = random.Random(seed)
    return [R.fq12_from_flat_ark([rng.randrange(R.P) for _ in range(12)]) for _ in range(n)]
