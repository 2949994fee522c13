"""A/B timing on one box: alternates processes over variants, reports the median kernel times.
usage: gpu_ab.py ROUNDS variant...   variant = name[:ENV=VAL[,ENV=VAL...]]; ENV BLS381_LIB_PATH may name another
build of the library (relative to the package directory)."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, _, envs = spec.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    if "BLS381_LIB_PATH" in env:
        env["BLS381_LIB_PATH"] = os.path.join(ROOT, "plonky2_bls12_381_pairing_amd", env["BLS381_LIB_PATH"])
    variants.append((name, env))
child = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from helpers import *
import plonky2_bls12_381_pairing_amd as B
ps, qs = random_pairs(6, 3)
n = int(os.environ.get("AB_N", "65536"))
g1 = np.tile(g1_array(ps), (n // 6 + 1, 1))[:n]; g2 = np.tile(g2_array(qs), (n // 6 + 1, 1))[:n]
ctx = B.Context(0, int(os.environ.get("AB_K", "0")), int(os.environ.get("AB_SLOTS", "0")))
for rep in range(6):
    ctx.pairing_batch(g1, g2); ms = ctx.last_timing()
    if rep: print("T", ms[0], ms[2], flush=True)
''' % (ROOT, ROOT)
res = {l: ([], []) for l, _ in variants}
for r in range(rounds):
    for l, venv in variants:
        env = dict(os.environ, BLS381_SLICE="65536"); env.update(venv)
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=200).stdout
        for line in out.splitlines():
            if line.startswith("T "):
                _, m, f = line.split()
                res[l][0].append(float(m)); res[l][1].append(float(f))
for l, _ in variants:
    m, f = res[l]
    print("%-24s miller median %.3f min %.3f | fe2 median %.3f min %.3f  (%d samples)" % (l, statistics.median(m), min(m), statistics.median(f), min(f), len(m)), flush=True)
