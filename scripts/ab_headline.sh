#!/bin/bash
# A/B of the configs[1] headline: round-2 library (git worktree at 67fcafc, built here) against HEAD, same box, same session,
# alternating, the driver's own arguments.
cd /root/repo   # (needs: git worktree add _ab/r2 67fcafc && (cd _ab/r2 && python -c "from cppnumericalsolvers_amd import _build; _build.build()"))
mkdir -p gpurun_out
OUT=gpurun_out/r5_ab_headline.txt
: > $OUT
for rep in 1 2 3; do
  for side in r2 head; do
    if [ $side = r2 ]; then dir=/root/repo/_ab/r2; else dir=/root/repo; fi
    line=$(cd $dir && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-counters 2>/dev/null | tail -1)
    echo "$side rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value %.4g ms_per_step %.4f kernel_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))')" >> $OUT
  done
done
cat $OUT
