#!/bin/bash
# Witness-trace kernel: throughput and HBM traffic against the number of proofs per pass (tuning aid / evidence).
# usage: gpu_witness_traffic.sh  -> gpurun_out/wit_traffic/{timing.txt, pmc_<counter>_<ch>/}
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${OUTDIR:-wit_traffic}
mkdir -p $OUT
export N=${N:-16384}
: > $OUT/timing.txt
for lpp in ${LPPS:-16 32}; do
  for ch in ${CHS:-16384 8192 4096 2048 1024 512}; do
    BLS381_WIT_LPP=$lpp BLS381_WIT_CH=$ch timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/gpu_witness_dev.py 2>&1 | grep "proofs/s" | sed "s/^/lpp $lpp /" >> $OUT/timing.txt || exit 2
  done
done
cd /tmp && export TMPDIR=/tmp
for ch in ${PMC_CH:-16384 2048 512}; do
  export BLS381_WIT_CH=$ch
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${ctr}_$ch -- python3 $GRAFT_REPO_ROOT/tools/gpu_witness_dev.py > $OUT/pmc_${ctr}_$ch.log 2>&1 || exit 3
  done
done
python3 - "$OUT" ${PMC_CH:-16384 2048 512} <<'PY'
import csv, glob, sys
out = sys.argv[1]
for ch in sys.argv[2:]:
    tot = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        s = 0.0
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, ctr, ch), recursive=True):
            for row in csv.DictReader(open(f)):
                if "witness_kernel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    s += float(row["Counter_Value"])
        tot[ctr] = s
    print("ch %s witness_kernel FETCH_SIZE_KiB %.0f WRITE_SIZE_KiB %.0f (two calls of N proofs each)" % (ch, tot["FETCH_SIZE"], tot["WRITE_SIZE"]))
PY
