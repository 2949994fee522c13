"""Turns the newest outputs of tools/gpu_round_profile.sh (under gpurun_out/prof_round/) into the files kept
under profiles/: the bench line, the rocprofv3 kernel statistics and the HBM traffic / instruction counts of the
step's kernels.  FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE is doubled (16-byte-per-lane loads on
gfx950), the factor pack_kernel confirms in the same run: it reads exactly the bytes it writes."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"   # file prefix: the round the measurement belongs to


def newest(pattern):
    files = glob.glob(os.path.join(SRC, pattern))
    return max(files, key=os.path.getmtime)


def counters(path):
    """{kernel short name: {counter: total over the run's dispatches}} and the number of bench steps in the run (one
    Miller-loop dispatch each).  Kernels: the Miller loop (vm_kernel<8, 0>), the step programs of the final
    exponentiation's second part (vm_kernel<8, 1>: six dispatches per step on large batches), the squaring-chain kernel
    (five dispatches per step), the layout conversion (calibration)."""
    out, steps = {}, set()
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        short = "fe2_programs" if "vm_kernel<8, 1>" in name else "miller" if "vm_kernel<8, 0>" in name else \
            "chain" if "cyc_chain_kernel" in name else \
            "pack" if "pack_kernel" in name and "unpack" not in name and "tree" not in name else None
        if short:
            out.setdefault(short, {}).setdefault(row["Counter_Name"], 0.0)
            out[short][row["Counter_Name"]] += float(row["Counter_Value"])
            if short == "miller":
                steps.add(row["Dispatch_Id"])
            if short == "pack":
                out[short].setdefault("_dispatches", set()).add(row["Dispatch_Id"])
    n = max(1, len(steps))
    res = {}
    for k, d in out.items():
        div = len(d.pop("_dispatches")) if k == "pack" else n
        res[k] = {c: v / div for c, v in d.items()}
    return res


fetch = counters(newest("pmc_fetch/*/*_counter_collection.csv"))
write = counters(newest("pmc_write/*/*_counter_collection.csv"))
sq = counters(newest("pmc_sq/*/*_counter_collection.csv"))
res = {"note": __doc__.strip().replace("\n", " ") + "  All figures are per bench step (65,536 pairings): one Miller-loop launch, "
               "six program launches and five squaring-chain launches of the final exponentiation's second part."}
KERNELS = ("miller", "fe2_programs", "chain")
for k in KERNELS:
    res[k + "_fetch_bytes"] = int(2 * 1024 * fetch[k]["FETCH_SIZE"])
    res[k + "_write_bytes"] = int(1024 * write[k]["WRITE_SIZE"])
    res[k + "_hbm_bytes_per_step"] = res[k + "_fetch_bytes"] + res[k + "_write_bytes"]
res["miller_hbm_bytes_per_launch"] = res["miller_hbm_bytes_per_step"]
res["miller_algorithmic_bytes_per_launch"] = 65536 * (6 + 12) * 48
res["chain_algorithmic_bytes_per_step"] = 5 * 65536 * (12 + 72) * 48
res["pack_kernel_calibration_KiB"] = {"FETCH_SIZE": fetch["pack"]["FETCH_SIZE"], "WRITE_SIZE": write["pack"]["WRITE_SIZE"]}
# waves per launch: 8 pairings per wave in the interpreter, 64 elements per three waves in the chain kernel
waves = {"miller": 65536 // 8, "fe2_programs": 65536 // 8, "chain": 3 * 65536 // 64}
for label, ctr in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS")):
    res[label + "_instructions_per_wave_per_step"] = {k: sq[k][ctr] / waves[k] for k in KERNELS if ctr in sq.get(k, {})}
# VALU pipe occupancy from the same pass: SQ_WAVE_CYCLES counts resident waves in units of 4 cycles; the interpreter
# keeps 2 waves on each of the 1,024 SIMDs, the chain kernel 3.
res["valu_pipe_busy_share"] = {}
for k in KERNELS:
    if "SQ_WAVE_CYCLES" in sq.get(k, {}):
        per_simd = 3 if k == "chain" else 2
        kernel_cycles = sq[k]["SQ_WAVE_CYCLES"] * 4 / (1024 * per_simd)
        res["valu_pipe_busy_share"][k] = sq[k]["SQ_INSTS_VALU"] / 1024 * 4 / kernel_cycles
        res.setdefault("kernel_cycles_per_step", {})[k] = kernel_cycles
        res.setdefault("lds_bank_conflict_cycles_per_cu", {})[k] = sq[k]["SQ_LDS_BANK_CONFLICT"] / 256
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_roofline_traffic.json" % tag), "w"), indent=1)
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(ROOT, "profiles", "%s_bench.json" % tag))
shutil.copy(newest("stats/*/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag))
print(json.dumps(res, indent=1))
