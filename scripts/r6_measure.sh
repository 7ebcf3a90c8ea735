#!/bin/bash
# Round-6 measurement pass on the GPU box (through gpurun):
#   1. the tail-split A/B (scripts/tail_split_probe.py)                       -> gpurun_out/r6_ab_tail.txt
#   2. full bench lines (counters + both CPU legs) of the driver's command line and of the SURVEY 8(f) rows
#                                                                              -> gpurun_out/r6_bench_n1_<name>.json
#   3. the augmented-Lagrangian row (n = 64) with counters and the reference  -> gpurun_out/r6_bench_n1_f_al.json
#   4. rocprofv3 --kernel-trace --stats of the same commands                   -> gpurun_out/prof_r6/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python scripts/tail_split_probe.py > $OUT/r6_ab_tail.txt 2> $OUT/r6_ab_tail.err; echo "tail probe rc=$?"
bash scripts/bench_all.sh r6 "${R6_SPECS:-default f_hz f_bfgs f_second}"
timeout 900 python scripts/auglag_bench.py --batch 16384 --n 64 --cpu-sample 2048 --counters > $OUT/r6_bench_n1_f_al.json 2> $OUT/r6_bench_n1_f_al.err; echo "al rc=$?"
tail -c 600 $OUT/r6_bench_n1_f_al.err
PROFILE_WORKLOADS="${R6_PROFILE:-cfg2 f_hz f_bfgs f_second}" bash scripts/profile_gpu.sh r6 > $OUT/r6_profile.log 2>&1
mkdir -p $OUT/prof_r6 && cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r6/stats_f_al -o f_al -- python $ROOT/scripts/auglag_bench.py --batch 16384 --n 64 --steps 2 --warmup 1 --no-cpu > $OUT/prof_r6/stats_f_al.log 2>&1
cd $ROOT && python scripts/summarize_profiles.py r6 > $OUT/r6_summarize.log 2>&1; tail -5 $OUT/r6_summarize.log
cat $OUT/r6_ab_tail.txt
