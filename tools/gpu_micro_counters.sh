#!/bin/bash
# Instruction counters of single-kind step programs for two builds of the library (tuning aid).
# usage: gpu_micro_counters.sh PROGRAM...   -> gpurun_out/microctr/<build>_<program>.csv
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/microctr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for prog in "$@"; do
  for build in ${BUILDS:-new old}; do
    if [ $build = old ]; then export BLS381_LIB_PATH=$GRAFT_REPO_ROOT/plonky2_bls12_381_pairing_amd/libbls381_old.so; else unset BLS381_LIB_PATH; fi
    [ $build = old ] && [ ! -f "$BLS381_LIB_PATH" ] && continue
    timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $OUT/${build}_$prog -- python3 $GRAFT_REPO_ROOT/tools/gpu_micro_one.py $prog > $OUT/${build}_$prog.log 2>&1 || exit 3
  done
done
echo done
