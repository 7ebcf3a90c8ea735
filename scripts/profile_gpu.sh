#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes for the bench
# workloads.  Outputs land in gpurun_out/prof_<tag>/; scripts/summarize_profiles.py turns
# them into the committed summaries under profiles/.
#   usage: bash scripts/profile_gpu.sh <tag> [extra bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-counters $*"
for WL in ${PROFILE_WORKLOADS:-cfg2 cfg3 cfg4 cfg5}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o $WL -- $BENCH --workload $WL > $OUT/stats_$WL.log 2>&1
  # PMC passes: one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$WL -o $WL -- $BENCH --workload $WL > $OUT/pmc_fetch_$WL.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$WL -o $WL -- $BENCH --workload $WL > $OUT/pmc_write_$WL.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$WL -o $WL -- $BENCH --workload $WL > $OUT/pmc_sq_$WL.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$WL -o $WL -- $BENCH --workload $WL > $OUT/pmc_sq2_$WL.log 2>&1
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
