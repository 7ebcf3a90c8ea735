#!/bin/bash
set -u
mkdir -p gpurun_out/r2
run() { python bench.py --no-cpu-baseline --no-secondary --no-counters --steps 4 --warmup 1 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-40s %-30s %10.0f solves/s %8.3f ms' % ('$LIBTAG', '$*', d['value'], d['ms_per_step']))"; }
for round in 1 2; do
  LIBTAG=default; unset MI355_LBFGS_LIBRARY; run --workload cfg5; run --workload cfg4
  LIBTAG=contract-fast; export MI355_LBFGS_LIBRARY=$PWD/cppnumericalsolvers_amd/variants/lib_fmafast.so; run --workload cfg5; run --workload cfg4
done | tee gpurun_out/r2/ab_cfg45_contract.txt
