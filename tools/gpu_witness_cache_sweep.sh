#!/bin/bash
# Witness-trace kernel: proofs/s against the number of rows per proof kept in LDS (and lanes per proof).
# usage: gpu_witness_cache_sweep.sh -> gpurun_out/wit_cache_sweep.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/wit_cache_sweep.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $OUT
export N=${N:-16384}
for lpp in ${LPPS:-16}; do
  for rows in ${ROWS:-0 16 24 32 40 48 52 64}; do
    BLS381_WIT_LPP=$lpp BLS381_WIT_CACHE=$rows timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/gpu_witness_dev.py 2>&1 | grep "proofs/s\|checksum" | sed "s/^/lpp $lpp /" >> $OUT || exit 2
  done
done
cat $OUT
