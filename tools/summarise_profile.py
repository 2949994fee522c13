"""Turns the newest outputs of tools/gpu_round_profile.sh (under gpurun_out/prof_round/) into the files kept
under profiles/: the bench line, the rocprofv3 kernel statistics and the HBM traffic / instruction counts of the
two interpreter kernels.  FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE is doubled (16-byte-per-lane loads on
gfx950), the factor pack_kernel confirms in the same run: it reads exactly the bytes it writes."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"   # file prefix: the round the measurement belongs to


def newest(pattern):
    files = glob.glob(os.path.join(SRC, pattern))
    return max(files, key=os.path.getmtime)


def counters(path):
    """{kernel short name: {counter: [values per dispatch]}}"""
    out = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        short = "fe2" if "vm_kernel<8, 1>" in name else "miller" if "vm_kernel<8, 0>" in name else \
            "pack" if "pack_kernel" in name and "unpack" not in name and "tree" not in name else None
        if short:
            out.setdefault(short, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            out[short][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in d.items()} for k, d in out.items()}


def mean(xs):
    return sum(xs) / len(xs)


fetch = counters(newest("pmc_fetch/*/*_counter_collection.csv"))
write = counters(newest("pmc_write/*/*_counter_collection.csv"))
sq = counters(newest("pmc_sq/*/*_counter_collection.csv"))
res = {"note": __doc__.strip().replace("\n", " ")}
for k in ("fe2", "miller"):
    res[k + "_fetch_bytes"] = int(2 * 1024 * mean(fetch[k]["FETCH_SIZE"]))
    res[k + "_write_bytes"] = int(1024 * mean(write[k]["WRITE_SIZE"]))
res["fe2_hbm_bytes_per_launch"] = res["fe2_fetch_bytes"] + res["fe2_write_bytes"]
res["fe2_algorithmic_bytes_per_launch"] = 65536 * (21 + 12) * 48
res["pack_kernel_calibration_KiB"] = {"FETCH_SIZE": mean(fetch["pack"]["FETCH_SIZE"]), "WRITE_SIZE": mean(write["pack"]["WRITE_SIZE"])}
waves = 65536 // 8  # one wave per 8 pairings at 8 lanes each
for label, ctr in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS")):
    res[label + "_instructions_per_wave"] = {k: mean(sq[k][ctr]) / waves for k in ("miller", "fe2") if ctr in sq.get(k, {})}
# VALU pipe occupancy from the same pass: a wave64 VALU instruction holds its SIMD's pipe for at least 4 cycles;
# SQ_WAVE_CYCLES counts resident waves in units of 4 cycles, and a launch of this size keeps 2 waves on each of
# the 1,024 SIMDs, so the kernel's duration in cycles is SQ_WAVE_CYCLES * 4 / 2048.
res["valu_pipe_busy_share"] = {}
for k in ("miller", "fe2"):
    if "SQ_WAVE_CYCLES" in sq.get(k, {}):
        kernel_cycles = mean(sq[k]["SQ_WAVE_CYCLES"]) * 4 / 2048
        res["valu_pipe_busy_share"][k] = mean(sq[k]["SQ_INSTS_VALU"]) / 1024 * 4 / kernel_cycles
        res.setdefault("kernel_cycles", {})[k] = kernel_cycles
        res.setdefault("lds_bank_conflict_cycles_per_cu", {})[k] = mean(sq[k]["SQ_LDS_BANK_CONFLICT"]) / 256
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_roofline_traffic.json" % tag), "w"), indent=1)
shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(ROOT, "profiles", "%s_bench.json" % tag))
shutil.copy(newest("stats/*/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % tag))
print(json.dumps(res, indent=1))
