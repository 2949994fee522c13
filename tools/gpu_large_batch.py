"""Timing of large device-resident batches (the per-GPU shard of the 8M configuration): pairings per second.
usage: gpu_large_batch.py [N ...]   (BLS381_LAZY_PRODUCTS=0 keeps every chunk at 8 lanes per pairing)"""
import os, sys, time
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd._lib import lib, check
ps, qs = H.random_pairs(16, 77)
a, b = H.g1_array(ps), H.g2_array(qs)
ctx = B.Context(0)
dev = torch.device("cuda", 0)
for n in [int(x) for x in sys.argv[1:]] or [1 << 20]:
    i = np.arange(n)
    g1 = torch.from_numpy(a[i % 16].view(np.int64)).to(dev)
    g2 = torch.from_numpy(b[(i // 16) % 16].view(np.int64)).to(dev)
    out = torch.empty((n, 72), dtype=torch.int64, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        check(lib.bls381_pairing_batch_dev(ctx.handle, g1.data_ptr(), g2.data_ptr(), None, n, out.data_ptr()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    print("lazy %s: %d pairings %.1f ms = %.3f M pairings/s, checksum %d" % (os.environ.get("BLS381_LAZY_PRODUCTS", "default"), n, best * 1e3, n / best / 1e6,
                                                                          int(out[::4099].sum().item())), flush=True)
