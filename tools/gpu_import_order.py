"""Probe: this package first, a context and a pairing, then torch on the same GPU (one HIP runtime for both)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
ps, qs = H.random_pairs(2, 1)
with B.Context(0) as ctx:
    a = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
    import torch
    assert torch.cuda.is_available(), "torch lost its GPU"
    t = torch.arange(8, device="cuda").sum().item()
    b = ctx.pairing_batch(H.g1_array(ps), H.g2_array(qs))
print("runtimes", sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)), "torch sum", t, "same", np.array_equal(a, b))
