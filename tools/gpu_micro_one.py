"""Runs one synthetic single-kind program (or a real one) a few times: the workload for counter passes.
usage: gpu_micro_one.py PROGRAM [K]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd import _lib

which = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = B.Context(0, K, 0)
out = (C.c_uint64 * 32)()
for _ in range(2):
    _lib.check(_lib.lib.bls381_debug_profile(ctx.handle, which.encode(), 65536, out))
print(which, "kernel ms", out[15] / 1e6)
ctx.close()
