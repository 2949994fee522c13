#!/bin/bash
# Witness-trace kernel: instruction and wait counters of one call of N proofs (where its time goes).
# usage: gpu_witness_counters.sh -> gpurun_out/wit_counters/{pmc/, summary.txt}
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/wit_counters
mkdir -p $OUT
export N=${N:-16384}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/gpu_witness_dev.py > $OUT/pmc.log 2>&1 || exit 3
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "witness_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot): print("%-22s %.4g" % (k, tot[k]))
w = tot["SQ_WAVES"]
if w:
    print("per wave: VALU %.0f SALU %.0f LDS %.0f; wave cycles %.4g; VALU busy / wave cycles %.3f; waiting / wave cycles %.3f" % (
        tot["SQ_INSTS_VALU"] / w, tot["SQ_INSTS_SALU"] / w, tot["SQ_INSTS_LDS"] / w, tot["SQ_WAVE_CYCLES"] / w,
        tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1), tot["SQ_WAIT_INST_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1)))
PY
