#!/bin/bash
# Per-kind instruction mix of the interpreter's steps (tuning aid): counters of single-kind step programs.
# usage: gpu_step_counters.sh PROGRAM...   -> gpurun_out/stepctr/<program>/ (rocprofv3 csv)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${OUTDIR:-stepctr}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for prog in "$@"; do
  timeout -s KILL 90 rocprofv3 --pmc ${COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_LDS} --output-format csv -d $OUT/$prog -- python3 $GRAFT_REPO_ROOT/tools/gpu_micro_one.py $prog > $OUT/$prog.log 2>&1 || exit 3
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for prog in sys.argv[2:]:
    acc = {}
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, prog), recursive=True):
        for row in csv.DictReader(open(f)):
            if "vm_kernel" not in row["Kernel_Name"]: continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(prog, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
