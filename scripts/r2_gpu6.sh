#!/bin/bash
# wide-lane (eight coordinates per lane) kernels: parity, then A/B against the default mapping on one box
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_fma.py -x -q -m gpu -k "eight_coordinates" > gpurun_out/r2/pytest_e8.log 2>&1
echo "e8 tests rc=$?"; tail -15 gpurun_out/r2/pytest_e8.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
run() { python bench.py --no-cpu-baseline --no-secondary --no-counters --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-60s %10.0f solves/s %8.3f ms  W%d E%d grid %s lds %s' % ('$*', d['value'], d['ms_per_step'], c['lanes_per_problem'], c['elems_per_lane'], c.get('grid_workgroups'), c.get('lds_bytes_per_workgroup')))"; }
for round in 1 2; do
  run --workload cfg2
  run --workload cfg2 --lanes 4 --elems 8
  run --workload cfg2 --batch 262144
  run --workload cfg2 --batch 262144 --lanes 4 --elems 8
  run --workload cfg3
  run --workload cfg3 --lanes 8 --elems 8
  run --workload cfg3full --steps 2 --warmup 1
  run --workload cfg3full --steps 2 --warmup 1 --lanes 8 --elems 8
done | tee gpurun_out/r2/ab_e8.txt
