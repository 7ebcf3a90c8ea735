#!/bin/bash
# instruction-cache counters of the solve kernels (config 5's kernel is 80 KB of code against a 64 KB cache shared by two CUs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WL in cfg5 cfg3 cfg4; do
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$WL -o $WL -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-counters --workload $WL > $OUT/$WL.log 2>&1
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$OUT/$WL/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "solve_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("$WL", {k: agg[k]/max(1,n[k]) for k in sorted(agg)})
PY
done
