#!/usr/bin/env python3
"""Per-step-kind cycle accounting of the interpreter programs on the GPU (tuning aid).
usage: gpu_profile.py [K:slots ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd import _lib

NAMES = {0: "product", 1: "sum", 2: "sum+sweep", 3: "two-product", 4: "copy in", 5: "copy fill", 6: "copy const",
         7: "copy spill", 8: "copy out", 9: "sum+reduce", 10: "summed-operand product", 11: "n-product"}
for spec in sys.argv[1:] or ["8:0"]:
    K, slots = (int(x) for x in spec.split(":"))
    ctx = B.Context(0, K, slots)
    progs = ("miller", "fe2")
    if os.environ.get("MICRO"):
        progs = os.environ["MICRO"].split(",") if "t_" in os.environ["MICRO"] else ("t_lin1", "t_lin2", "t_lin4", "t_lin7", "t_sweep4", "t_reduce4", "t_mul", "t_dot2", "t_dot2w2", "t_dot2w3", "t_spill", "t_fill", "t_const")
    for which in progs:
        out = (C.c_uint64 * 32)()
        if _lib.lib.bls381_debug_profile(ctx.handle, which.encode(), 65536, out) != 0:
            print(which, "not available:", _lib.lib.bls381_last_error().decode())
            continue
        total = sum(out[:15])
        nst = sum(out[16:])
        waves = 65536 * K // 64
        print("K=%d slots=%d %s: %d steps, %.0f cycles, kernel %.3f ms = %.1f ns/step/wave-slot" % (
            K, slots, which, nst, total, out[15] / 1e6, out[15] / nst / (waves / (256 * 2 * 4.0))))
        for k in range(16):
            if out[16 + k]:
                print("   %-12s steps %5d  cycles/step %7.0f  share %5.1f%%" % (NAMES.get(k, k), out[16 + k], out[k] / out[16 + k],
                                                                               100.0 * out[k] / total))
    ctx.close()
