"""Probe (tuning aid): one batch of 65,536 pairings as S concurrent shards on S contexts of one GPU, against one context."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
import plonky2_bls12_381_pairing_amd as B
from plonky2_bls12_381_pairing_amd import _lib
N = 65536
ps, qs = H.random_pairs(6, 3)
g1 = np.tile(H.g1_array(ps), (N // 6 + 1, 1))[:N]; g2 = np.tile(H.g2_array(qs), (N // 6 + 1, 1))[:N]
dev = torch.device("cuda", 0)
d_g1 = torch.from_numpy(g1.view(np.int64)).to(dev); d_g2 = torch.from_numpy(g2.view(np.int64)).to(dev)
d_out = torch.empty((N, 72), dtype=torch.int64, device=dev)
ref = None
for spec in sys.argv[1:]:
    K, S = (int(x) for x in spec.split(":"))
    ctxs = [B.Context(0, K, 0) for _ in range(S)]
    bounds = [N * i // S for i in range(S + 1)]
    def run(i):
        lo, hi = bounds[i], bounds[i + 1]
        _lib.check(_lib.lib.bls381_pairing_batch_dev(ctxs[i].handle, d_g1.data_ptr() + lo * 96, d_g2.data_ptr() + lo * 192, None, hi - lo,
                                                     d_out.data_ptr() + lo * 576))
    def step():
        if S == 1:
            run(0); return
        th = [threading.Thread(target=run, args=(i,)) for i in range(S)]
        for t in th: t.start()
        for t in th: t.join()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    out = d_out.cpu().numpy()
    if ref is None: ref = out.copy()
    print("K=%d shards=%d: %.2f ms per 65,536 (%.2f M/s) equal=%s" % (K, S, ms, N / ms / 1e3, np.array_equal(out, ref)), flush=True)
    for c in ctxs: c.close()
