#!/usr/bin/env python3
"""Headline benchmark: BLS12-381 pairings per second on MI355X.

One step = one batch of 65,536 independent pairings (Miller loop + final exponentiation,
BASELINE.json configs[1]) per GPU, inputs and outputs resident in HBM, through the C ABI's
device-pointer entry point.  With N > 1 ranks (launched by torch.distributed.run, one process per
GPU) every rank runs its own batch: independent pairings shard with no collective ("weak" scaling);
the process group is only used for the timing barrier and the MAX reduction of the elapsed time.

Prints ONE JSON line on rank 0 with the fields the driver expects plus
  roofline      dominant kernel (final exponentiation part 2) against the v_mad_u64_u32 issue rate
  cpu_baseline  the C restatement of the reference's native path (oracle/c) on the host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))

BATCH = 65536
# Algorithmic work: Fp-multiplication-equivalents of the reference's formulas x 300 multiply-accumulates
# (12 x 32-bit limbs: 144 for the product, 156 for the Montgomery reduction)
W_PAIR = 15260 * 300          # whole pairing
W_MILLER = 7141 * 300         # G2 line precompute + Miller loop
W_FE2 = (8119 - 460 - 75) * 300   # final exponentiation minus the Fp inversion chain and the norm part
PEAK_MAC = 256 * 4 * 64 / 4 * 2.4e9   # v_mad_u64_u32: 4 cycles per wave64 instruction per SIMD (measured), 2.4 GHz


def synth_inputs(np, H, n):
    """256 distinct G1 x 256 distinct G2 points from seeded scalars -> n distinct (P, Q) pairs."""
    import random
    rng = random.Random(0x5EEDB15381)
    m = 256
    ps = [H.R.g1_mul(H.R.G1_GEN, rng.randrange(1, H.R.R_ORDER)) for _ in range(m)]
    qs = [H.R.g2_mul(H.R.G2_GEN, rng.randrange(1, H.R.R_ORDER)) for _ in range(m)]
    g1b, g2b = H.g1_array(ps), H.g2_array(qs)
    i = np.arange(n)
    return g1b[i % m], g2b[(i // m) % m], g1b, g2b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per pairing (0 = library default)")
    ap.add_argument("--lds-slots", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import helpers as H
    import plonky2_bls12_381_pairing_amd as B
    from plonky2_bls12_381_pairing_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    dist = None
    # Rehearsal of the multi-rank control flow on a box with one GPU: BLS381_BENCH_BACKEND=gloo and
    # BLS381_BENCH_DEVICE=0 put every rank on the same card (RCCL refuses two ranks on one device).
    backend = os.environ.get("BLS381_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("BLS381_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    # BLS381_BENCH_FORCE_DIST=1 builds the process group even for one rank: on a one-GPU box that runs the RCCL
    # code path (initialisation, barrier, device-tensor all-reduce and all-gather) the multi-GPU launch will take
    if world > 1 or os.environ.get("BLS381_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ctx = B.Context(device_index, args.lanes, args.lds_slots)
    g1, g2, g1b, g2b = synth_inputs(np, H, BATCH)
    dev = torch.device("cuda", device_index)
    d_g1 = torch.from_numpy(g1.view(np.int64)).to(dev)
    d_g2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    d_out = torch.empty((BATCH, 72), dtype=torch.int64, device=dev)

    def step():
        _lib.check(_lib.lib.bls381_pairing_batch_dev(ctx.handle, d_g1.data_ptr(), d_g2.data_ptr(), None, BATCH,
                                                     d_out.data_ptr()))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        step()                       # returns after the context's stream has drained
        kern_ms.append(ctx.last_timing())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness outside the timed region: a strided sample against the CPU oracle, bit for bit
    out = d_out.cpu().numpy().view(np.uint64)
    import c_oracle as CO
    sample = np.arange(0, BATCH, BATCH // 16) + rank
    ref = CO.pairing_batch(g1[sample], g2[sample])
    if not np.array_equal(out[sample], ref):
        raise SystemExit("rank %d: GPU pairings differ from the CPU oracle" % rank)

    # which ranks took part: every rank contributes its number through the process group
    ranks_seen = [0]
    if dist is not None:
        mine = torch.tensor([rank], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        ranks_seen = sorted(int(t.item()) for t in got)

    if rank == 0:
        value = world * BATCH * args.steps / elapsed
        ms_fe2 = float(np.mean([k[2] for k in kern_ms]))
        ms_miller = float(np.mean([k[0] for k in kern_ms]))
        ms_fe1 = float(np.mean([k[1] for k in kern_ms]))
        achieved = BATCH * W_FE2 / (ms_fe2 * 1e-3)
        peak_measured = ctx.calibrate_mac_rate()
        # the clock the calibration loop ran at: its rate over one multiply-accumulate per lane per 4 cycles
        calib_clock_ghz = peak_measured / (256 * 4 * 64 / 4) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("fe2_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "BLS12-381 pairings/sec (node)",
            "value": value,
            "unit": "pairings/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "backend": backend if dist is not None else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "integer: 14 signed 28-bit digits per Fp value in u32 (64-bit column accumulators); 12 x u32 "
                     "Montgomery limbs at the boundary",
            "data": "synthetic: 256 seeded G1 x 256 seeded G2 multiples of the generators -> 65,536 distinct pairs per GPU",
            "config": {"workload": "65,536 independent pairings (Miller loop + final exponentiation) per GPU per step, "
                                   "bit-exact vs the CPU oracle (BASELINE configs[1])",
                       "batch_per_gpu": BATCH, "lanes_per_pairing": args.lanes or 8, "parallelism": "independent shards, no collective"},
            "roofline": {
                "bound": "valu_int_mul",
                "kernel": "vm_kernel<K, TAG_FINAL_EXP> (final exponentiation part 2)",
                "achieved": achieved / 1e12,
                "peak": PEAK_MAC / 1e12,
                "unit": "TMAC/s (32x32->64 multiply-accumulates)",
                "frac": achieved / PEAK_MAC,
                "traffic": traffic,
                "traffic_source": "profiles/roofline_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command recorded earlier (tools/gpu_round_profile.sh), not measured in this run",
                "peak_measured": peak_measured / 1e12,
                "peak_measured_clock_GHz": calib_clock_ghz,
                "frac_of_measured_peak": achieved / peak_measured,
                "whole_step_frac": value / world * W_PAIR / PEAK_MAC,
                "kernel_ms": {"miller": ms_miller, "fe1_plus_inversion": ms_fe1, "fe2": ms_fe2},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # every core this process may use: its affinity mask, or what the box grants through OMP_NUM_THREADS
            # (a one-GPU share of a larger host); the machine's total is stated beside it
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            granted = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            cores = max(1, min(avail, granted) if granted > 0 else avail)
            res["host_cores_total"] = os.cpu_count()
            CO.set_threads(cores)
            probe = np.arange(8 * cores) * 37 % BATCH
            t1 = time.perf_counter()
            CO.pairing_batch(g1[probe], g2[probe])                  # warms the threads, sizes the sample
            rate = len(probe) / (time.perf_counter() - t1)
            nsamp = int(min(BATCH, max(len(probe), rate * 22.0)))   # 10-20 s of CPU work (the cold probe reads low)
            idx = np.arange(nsamp) * 37 % BATCH
            t1 = time.perf_counter()
            CO.pairing_batch(g1[idx], g2[idx])
            dt = time.perf_counter() - t1
            # second row: the same work with an optimised final exponentiation (cyclotomic squarings), so that
            # the reference's slow MyFq12 products do not inflate the ratio
            CO.set_fast(True)
            t2 = time.perf_counter()
            CO.pairing_batch(g1[idx], g2[idx])
            dt_fast = time.perf_counter() - t2
            CO.set_fast(False)
            # third figure: the reference's algorithm on one host thread (a short sample: about 2 s)
            CO.set_threads(1)
            n_one = int(min(nsamp, max(16, nsamp / dt / cores * 2.0)))
            t3 = time.perf_counter()
            CO.pairing_batch(g1[idx[:n_one]], g2[idx[:n_one]])
            dt_one = time.perf_counter() - t3
            CO.set_threads(cores)
            res["cpu_baseline_one_thread"] = {
                "value": n_one / dt_one, "unit": "pairings/s", "cores": 1, "kind": "port",
                "sample": "the first %d of those pairings, the reference's algorithm, one thread, %.1f s" % (n_one, dt_one),
            }
            res["cpu_baseline_optimised"] = {
                "value": nsamp / dt_fast, "unit": "pairings/s", "cores": cores, "kind": "port",
                "sample": "the same %d pairings with Granger-Scott cyclotomic squarings and Karatsuba tower products "
                          "in the final exponentiation (same values, not the reference's algorithm), %.1f s" % (nsamp, dt_fast),
            }
            res["cpu_baseline"] = {
                "value": nsamp / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
                "sample": "%d pairings of the same batch (C restatement of the reference's native algorithm incl. its "
                          "144-multiplication MyFq12 final exponentiation; Frobenius constants precomputed), "
                          "OpenMP over %d host threads, %.1f s" % (nsamp, cores, dt),
            }
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
