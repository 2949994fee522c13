#!/bin/bash
# Witness-trace kernel: proofs/s against the level assignment (BLS381_WIT_GROUP: 0 = every row at its earliest
# level, g = rows held back to fill groups of g), lanes per proof and rows per proof kept in LDS.
# CFGS="group:lanes:cache ..." chooses the points.
# usage: gpu_witness_group_sweep.sh -> gpurun_out/wit_group_sweep.txt
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/wit_group_sweep.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $OUT
export N=${N:-16384}
for cfg in ${CFGS:-0:16:48 16:16:48 16:16:32 16:16:64 8:8:48 32:32:48 32:16:48 16:32:48}; do
  set -- ${cfg//:/ }
  BLS381_WIT_GROUP=$1 BLS381_WIT_LPP=$2 BLS381_WIT_CACHE=$3 timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/gpu_witness_dev.py 2>&1 | grep "proofs/s\|checksum" | sed "s/^/group $1 lpp $2 cache $3: /" >> $OUT || exit 2
done
cat $OUT
