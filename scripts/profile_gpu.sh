#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + PMC passes for the bench workloads.  Outputs land in
# gpurun_out/prof_<tag>/; scripts/summarize_profiles.py turns them into the committed summaries under profiles/.
#   usage: bash scripts/profile_gpu.sh <tag>
# PROFILE_WORKLOADS: space-separated specs  name[:bench-arg[:bench-arg...]]  (the name keys the output directory; the
# part of the name before the first "_" is the bench workload).  Default: every BASELINE configuration with the
# kernel bench.py times by default, plus the alternative kernels of configs[3] / configs[4] and the whole configs[2]
# batch on one GPU.
set -u
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPECS=${PROFILE_WORKLOADS:-"cfg2 cfg3 cfg3full cfg4 cfg4_mfma:--ridge-mfma cfg4big cfg4own cfg5 cfg5_exact:--arithmetic:exact wide"}
for SPEC in $SPECS; do
  NAME=${SPEC%%:*}
  WL=${NAME%%_*}
  # the SURVEY 8(f) rows: f_hz, f_bfgs, f_second (+ a suffix after the second "_": f_hz_exact = f_hz with extra arguments)
  [[ "$NAME" == f_* ]] && WL=$(echo "$NAME" | cut -d_ -f1,2)
  EXTRA=""
  [[ "$SPEC" == *:* ]] && EXTRA=$(echo "${SPEC#*:}" | tr ':' ' ')
  BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-counters --workload $WL $EXTRA"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NAME -o $NAME -- $BENCH > $OUT/stats_$NAME.log 2>&1
  # PMC passes: one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass); never together with a trace
  # domain other than --kernel-trace
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$NAME -o $NAME -- $BENCH > $OUT/pmc_fetch_$NAME.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$NAME -o $NAME -- $BENCH > $OUT/pmc_write_$NAME.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq_$NAME -o $NAME -- $BENCH > $OUT/pmc_sq_$NAME.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$NAME -o $NAME -- $BENCH > $OUT/pmc_sq2_$NAME.log 2>&1
  # fp64 instruction classes (executed lane-flops = 64 (2 FMA + ADD + MUL + TRANS) + 512 MFMA_MOPS): solve kernel + pre-pass
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $OUT/pmc_flops_$NAME -o $NAME -- $BENCH > $OUT/pmc_flops_$NAME.log 2>&1
done
find $OUT -name "*.csv" | wc -l
du -sh $OUT
