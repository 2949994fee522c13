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
  verified_outputs   how many of the step's 65,536 outputs were compared with the CPU oracle's (every pairing the
                     baseline leg computes is compared, none is thrown away)
and, outside the timed region (rank 0, one GPU; --no-extras leaves them out), the other configured workloads:
  multi_pairing_513  the 512-key aggregate check (513 pairs -> one Fq12 -> final exponentiation -> is one), valid and
                     forged, latency per call
  signature_checks_65536  65,536 signature checks from compressed bytes (decoders, subgroup tests, two-pair pairing),
                     with and without the subgroup tests; one signature in eight forged
  shard_1m           one GPU's shard of the 8,388,608-pair configuration: 1,048,576 pairings on device pointers
  witness            the pairing circuit's witness trace: 256 proofs through host pointers and 16,384 proofs left in
                     HBM, proofs/s, fractions of the multiply-accumulate and HBM rooflines, every row of one proof
                     against the CPU replay of the circuit
With N > 1 ranks (or BLS381_BENCH_FORCE_DIST=1) the 513-pair product is also run sharded over the ranks: each rank
reduces its share to one Fq12, the partials are all-gathered through the process group (RCCL over xGMI) and every
rank finishes and checks the result (multi_pairing_rccl).
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
# what a pairing call executes of that model: its Miller loop runs the G2 doublings without their halvings, four
# multiplications fewer in each of the 63 doublings (same pairing values)
W_PAIR_EXECUTED = (15260 - 63 * 4) * 300
W_MILLER_EXECUTED = (7141 - 63 * 4) * 300   # the Miller loop as pairing calls run it
W_CHAIN = 63 * 18 * 300       # one run of the squaring-chain kernel: 63 cyclotomic squarings of 18 Fp multiplications
W_WITNESS = 70656 * 300       # the trace's mul rows
WITNESS_BYTES = 141364 * 48   # bytes of trace rows written per proof (operation rows)
PEAK_HBM = 8e12
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


def aggregate_check_inputs(H):
    """512 (public key, message point) pairs and (-G1, signature): the product of the 513 pairings is one.  Eight
    distinct keys and message points, tiled.  Returns g1 (513, 12), g2 (513, 24), and g2 with a forged signature."""
    import random
    R = H.R
    rng = random.Random(512)
    n, base = 512, 8
    sks = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
    hs = [rng.randrange(1, R.R_ORDER) for _ in range(base)]
    pks = [R.g1_mul(R.G1_GEN, s) for s in sks]
    Hs = [R.g2_mul(R.G2_GEN, h) for h in hs]
    ps = [pks[i % base] for i in range(n)] + [R.g1_neg(R.G1_GEN)]
    qs = [Hs[i % base] for i in range(n)]
    sigma = sum(sks[i % base] * hs[i % base] for i in range(n)) % R.R_ORDER
    good = qs + [R.g2_mul(R.G2_GEN, sigma)]
    forged = qs + [R.g2_mul(R.G2_GEN, (sigma + 1) % R.R_ORDER)]
    return H.g1_array(ps), H.g2_array(good), H.g2_array(forged)


def bench_multi_pairing(np, H, ctx):
    g1, g2, g2_forged = aggregate_check_inputs(H)
    e, one = ctx.native_pairing(g1, g2)
    _, one_forged = ctx.native_pairing(g1, g2_forged)
    if not one or one_forged or not np.array_equal(e, H.fq12_array([H.R.FQ12_ONE])[0]):
        raise SystemExit("the 513-pair aggregate check gives the wrong verdict")
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.native_pairing(g1, g2)
    return {"pairs": 513, "ms_per_check": (time.perf_counter() - t0) / reps * 1e3, "is_one": True, "forged_is_one": False,
            "what": "bls381_multi_pairing on host pointers: 513 Miller loops, in-kernel product tree, one final exponentiation"}


def bench_signature_checks(np, H, ctx):
    """65,536 signature checks from compressed bytes (bls381_verify_batch: three decoders, the subgroup tests as step
    programs, the two-pair pairing, a verdict byte per item).  Eight distinct (key, message point, signature) triples
    from the Python oracle, tiled; one triple in eight carries a forged signature."""
    import random
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import bls381_points as PT
    R = H.R
    rng = random.Random(65536)
    pk, mg, sg = [], [], []
    for i in range(8):
        sk, h = rng.randrange(1, R.R_ORDER), rng.randrange(1, R.R_ORDER)
        hm = R.g2_mul(R.G2_GEN, h)
        pk.append(PT.g1_compress(R.g1_mul(R.G1_GEN, sk)))
        mg.append(PT.g2_compress(hm))
        sg.append(PT.g2_compress(R.g2_mul(hm, sk if i != 5 else sk + 1)))
    n = 65536
    tile = lambda enc, w: np.tile(np.frombuffer(b"".join(enc), dtype=np.uint8).reshape(8, w), (n // 8, 1))
    bpk, bmg, bsg = tile(pk, 48), tile(mg, 96), tile(sg, 96)
    out = {"items": n, "what": "bls381_verify_batch on host pointers: 48 + 96 + 96 bytes in, one verdict byte out per item"}
    for label, mask in (("all_subgroup_tests", 7), ("no_subgroup_tests", 0)):
        verdict, _ = ctx.verify_batch(bpk, bmg, bsg, check_subgroup=mask)
        if verdict.tolist() != [0 if i % 8 == 5 else 1 for i in range(n)]:
            raise SystemExit("signature checks from bytes give wrong verdicts")
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.verify_batch(bpk, bmg, bsg, check_subgroup=mask)
            best = min(best, time.perf_counter() - t0)
        out[label] = {"ms": best * 1e3, "checks_per_s": n / best}
    out["verdicts"] = "57,344 valid and 8,192 forged signatures told apart in both runs"
    return out


def bench_shard(np, torch, _lib, ctx, dev, g1, g2, d_out_batch):
    """1,048,576 pairings on device pointers: the 65,536 verified pairs of the timed step, sixteen times over."""
    n = 1 << 20
    reps = n // BATCH
    d1 = torch.from_numpy(np.tile(g1, (reps, 1)).view(np.int64)).to(dev)
    d2 = torch.from_numpy(np.tile(g2, (reps, 1)).view(np.int64)).to(dev)
    out = torch.empty((n, 72), dtype=torch.int64, device=dev)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib.bls381_pairing_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), None, n, out.data_ptr()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same = all(bool(torch.equal(out[k * BATCH:(k + 1) * BATCH], d_out_batch)) for k in range(reps))
    if not same:
        raise SystemExit("the 1,048,576-pair shard differs from the verified 65,536-pair batch it repeats")
    return {"pairings": n, "ms": best * 1e3, "pairings_per_s": n / best, "whole_step_frac": n / best * W_PAIR_EXECUTED / PEAK_MAC,
            "outputs_equal_to_the_verified_batch": n}


def bench_witness(np, torch, _lib, H, B, ctx, dev):
    import witness_ref as W
    R = H.R
    ps, qs = H.random_pairs(8, 31)
    a, b = H.g1_array(ps), H.g2_array(qs)
    rows = B.witness_rows()
    # every row of proof 0 from the CPU replay of the circuit's builder calls
    tr, _ = W.pairing_witness(ps[0], qs[0])
    want = np.array([R.fq_to_u32_limbs(v) for _, v in tr.rows], dtype=np.uint32)
    res = {"rows_per_proof": rows, "bytes_per_proof": rows * 48}
    # host pointers, the configured 256 proofs (the copy of 6.79 MB per proof back to the host is most of it)
    n = 256
    g1, g2 = np.tile(a, (n // 8, 1)), np.tile(b, (n // 8, 1))
    ctx.pairing_witness_batch(g1[:8], g2[:8])
    t0 = time.perf_counter()
    trace = ctx.pairing_witness_batch(g1, g2)
    dt = time.perf_counter() - t0
    if not (np.array_equal(trace[0], want) and np.array_equal(trace[8], want)):
        raise SystemExit("witness trace (host form) differs from the CPU replay of the circuit")
    res["host_256"] = {"proofs": n, "ms": dt * 1e3, "proofs_per_s": n / dt, "rows_compared_with_cpu_replay": 2 * rows,
                       "output": "a fresh 1.7 GB array: its first touch (page faults) is inside the call"}
    # the same call writing into that array again, as a caller that keeps its buffer does
    trace[0] = 0; trace[n - 8] = 0
    t0 = time.perf_counter()
    ctx.pairing_witness_batch(g1, g2, out=trace)
    dt = time.perf_counter() - t0
    if not (np.array_equal(trace[0], want) and np.array_equal(trace[n - 8], want)):
        raise SystemExit("witness trace (host form, reused buffer) differs from the CPU replay of the circuit")
    res["host_256_reused_buffer"] = {"proofs": n, "ms": dt * 1e3, "proofs_per_s": n / dt}
    del trace
    # the trace left in HBM: 16,384 proofs in one call (the library takes them in passes of 8,192)
    n = 16384
    free = torch.cuda.mem_get_info()[0]
    while n > 1024 and n * rows * 48 > 0.8 * free:
        n //= 2
    d1 = torch.from_numpy(np.tile(a, (n // 8, 1)).view(np.int64)).to(dev)
    d2 = torch.from_numpy(np.tile(b, (n // 8, 1)).view(np.int64)).to(dev)
    d_trace = torch.empty((n, rows, 12), dtype=torch.int32, device=dev)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib.bls381_pairing_witness_batch_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), n, d_trace.data_ptr()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    first = d_trace[0].cpu().numpy().view(np.uint32)
    last = d_trace[n - 8].cpu().numpy().view(np.uint32)
    if not (np.array_equal(first, want) and np.array_equal(last, want)):
        raise SystemExit("witness trace (device form) differs from the CPU replay of the circuit")
    # a checksum over all proofs: every group of eight proofs repeats the first eight
    groups = d_trace.view(n // 8, 8 * rows * 12)
    sums = torch.cat([groups[k:k + 32].sum(dim=1, dtype=torch.int64) for k in range(0, n // 8, 32)])   # (in slices: the widened copy stays small)
    if not bool((sums == sums[0]).all()):
        raise SystemExit("witness trace checksums differ between identical proofs")
    rate = n / best
    # memory traffic per proof by the PMC counters, recorded with tools/gpu_witness_traffic.sh (not measured in this run)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r3_witness_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))["round3_final_kernel"]["hbm_bytes_per_proof"]
        except Exception:
            traffic = None
    res["hbm_bytes_per_proof"] = traffic
    res["hbm_bytes_source"] = "profiles/r3_witness_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes recorded earlier"
    res["device_%d" % n] = {"proofs": n, "ms": best * 1e3, "proofs_per_s": rate,
                            "mac_roofline_frac": rate * W_WITNESS / PEAK_MAC, "hbm_roofline_frac": rate * WITNESS_BYTES / PEAK_HBM,
                            "trace_GB_per_s_written": rate * rows * 48 / 1e9, "rows_compared_with_cpu_replay": 2 * rows,
                            "checksum_groups_equal": n // 8}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per pairing (0 = library default)")
    ap.add_argument("--lds-slots", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="leave out the workloads measured outside the timed region")
    args = ap.parse_args()

    # The JSON line is the only thing this program writes to its standard output: native libraries (RCCL prints a
    # version banner when a process group comes up) get the standard error stream instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
        if world == 1:  # (a forced one-rank group started without a launcher: rendezvous with itself)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
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
    kern_ms, chain_ms = [], []
    for _ in range(args.steps):
        step()                       # returns after the context's stream has drained
        kern_ms.append(ctx.last_timing())
        chain_ms.append(ctx.last_chain_timing())
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
    verified = set(int(i) for i in sample)

    # The 513-pair product (512 keys and their message points, and the signature against -G1) sharded over the ranks:
    # each rank reduces its share to one Fq12 on its GPU, one all-gather of 72 u64 per rank, every rank multiplies the
    # partials out, runs the one final exponentiation and compares with the same check done on its GPU alone.
    rccl = None
    if dist is not None:
        from plonky2_bls12_381_pairing_amd.parallel import all_gather_partials, shard_range
        mg1, mg2, mg2_forged = aggregate_check_inputs(H)
        alone, alone_one = ctx.native_pairing(mg1, mg2)
        lo, hi = shard_range(len(mg1), rank, world)
        gdev = dev if backend == "nccl" else None

        def sharded(q):
            part = ctx.multi_miller_loop_native(mg1[lo:hi], q[lo:hi])
            return ctx.finish_multi_pairing(all_gather_partials(part, device=gdev))
        e, one = sharded(mg2)
        e_forged, one_forged = sharded(mg2_forged)
        ok = bool(one) and bool(alone_one) and np.array_equal(e, alone) and not one_forged
        barrier()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            sharded(mg2)
        barrier()
        t_sh = torch.tensor([(time.perf_counter() - t1) / reps, 0.0 if ok else 1.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t_sh, op=dist.ReduceOp.MAX)
        if float(t_sh[1].item()) != 0.0:
            raise SystemExit("rank %d: the sharded 513-pair product differs from the one-GPU result" % rank)
        rccl = {"pairs": len(mg1), "ranks": world, "backend": backend, "ms_per_check": float(t_sh[0].item()) * 1e3,
                "is_one": bool(one), "forged_is_one": bool(one_forged), "equals_one_gpu_result": True,
                "exchange": "all_gather of one 576-byte Fq12 partial product per rank, then bls381_finish_multi_pairing on every rank"}

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
        # The dominant kernel of a step is the Miller loop (one launch).  The final exponentiation's second part is
        # eleven launches on large batches: five runs of the squaring-chain kernel and six step programs.
        achieved = BATCH * W_MILLER_EXECUTED / (ms_miller * 1e-3)
        achieved_fe2 = BATCH * W_FE2 / (ms_fe2 * 1e-3)
        chained = all(c is not None for c in chain_ms)
        ms_chain = float(np.mean([c[0] for c in chain_ms])) if chained else None
        ms_progs = float(np.mean([c[1] for c in chain_ms])) if chained else None
        peak_measured = ctx.calibrate_mac_rate()
        # the clock the calibration loop ran at: its rate over one multiply-accumulate per lane per 4 cycles
        calib_clock_ghz = peak_measured / (256 * 4 * 64 / 4) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("miller_hbm_bytes_per_launch")
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
                "kernel": "vm_kernel<K, TAG_MILLER> (the Miller loop incl. the G2 line computation: the longest-running kernel of a step)",
                "work_per_launch": "65,536 x (7,141 - 4 x 63) Fp multiplications x 300 multiply-accumulates: the reference's "
                                   "formulas less the halvings of the G2 doublings that pairing calls leave out",
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
                "whole_step_frac": value / world * W_PAIR_EXECUTED / PEAK_MAC,
                "whole_step_work": "%d multiply-accumulates per pairing: the reference's 15,260 Fp multiplications less the 4 "
                                   "per G2 doubling that pairing calls leave out, x 300" % W_PAIR_EXECUTED,
                "kernel_ms": {"miller": ms_miller, "fe1_plus_inversion": ms_fe1, "fe2": ms_fe2,
                              "fe2_chain_kernel_5_launches": ms_chain, "fe2_programs_6_launches": ms_progs},
            },
            # the other kernels of the step, same units and peak
            "roofline_final_exp_part2": {
                "what": "final exponentiation part 2 as a whole (five squaring-chain launches + six step programs)" if chained
                        else "vm_kernel<K, TAG_FINAL_EXP> (one program)",
                "achieved": achieved_fe2 / 1e12, "frac": achieved_fe2 / PEAK_MAC, "frac_of_measured_peak": achieved_fe2 / peak_measured,
            },
            "roofline_chain_kernel": None if not chained else {
                "kernel": "cyc_chain_kernel (63 cyclotomic squarings per element, element in registers)",
                "ms_per_launch": ms_chain / 5,
                "achieved": BATCH * W_CHAIN / (ms_chain / 5 * 1e-3) / 1e12, "frac": BATCH * W_CHAIN / (ms_chain / 5 * 1e-3) / PEAK_MAC,
            },
        }
        if rccl is not None:
            res["multi_pairing_rccl"] = rccl
        res["hip_runtime"] = _lib.hip_runtime_in_use()
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
            cpu_out = CO.pairing_batch(g1[idx], g2[idx])
            dt = time.perf_counter() - t1
            # second row: the same work with an optimised final exponentiation (cyclotomic squarings), so that
            # the reference's slow MyFq12 products do not inflate the ratio
            CO.set_fast(True)
            t2 = time.perf_counter()
            cpu_out_fast = CO.pairing_batch(g1[idx], g2[idx])
            dt_fast = time.perf_counter() - t2
            CO.set_fast(False)
            # every pairing the baseline computed is also a check of the GPU's output for the same pair
            if not (np.array_equal(out[idx], cpu_out) and np.array_equal(cpu_out_fast, cpu_out)):
                bad = np.nonzero((out[idx] != cpu_out).any(axis=1))[0]
                raise SystemExit("GPU pairings differ from the CPU oracle at %d of %d compared outputs (first: %s)" %
                                 (len(bad), nsamp, idx[bad[:4]].tolist()))
            verified.update(int(i) for i in idx)
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
        res["verified_outputs"] = {"compared": len(verified), "of": BATCH, "against": "CPU oracle (C restatement of the reference), bit for bit"}
        if world == 1 and not args.no_extras:
            t_extra = time.perf_counter()
            res["multi_pairing_513"] = bench_multi_pairing(np, H, ctx)
            res["signature_checks_65536"] = bench_signature_checks(np, H, ctx)
            res["shard_1m"] = bench_shard(np, torch, _lib, ctx, dev, g1, g2, d_out)
            res["witness"] = bench_witness(np, torch, _lib, H, B, ctx, dev)
            res["extras_s"] = time.perf_counter() - t_extra
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
