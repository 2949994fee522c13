#!/bin/bash
# Runs on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM traffic counters.
# Outputs under gpurun_out/prof_round/ (copy what should be kept into profiles/).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_round
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 280 python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1 || exit 4
echo done
