#!/bin/bash
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu -k "ridge or hessian" > gpurun_out/r2/pytest_ridge2.log 2>&1
echo "ridge tests rc=$?"; tail -12 gpurun_out/r2/pytest_ridge2.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
run() { python bench.py --no-cpu-baseline --no-secondary --no-counters --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-50s %10.0f solves/s %8.3f ms  W%d E%d grid %s threads %s lds %s mfma %.3f' % ('$*', d['value'], d['ms_per_step'], c['lanes_per_problem'], c['elems_per_lane'], c.get('grid_workgroups'), c.get('threads_per_workgroup'), c.get('lds_bytes_per_workgroup'), d.get('roofline_mfma',{}).get('frac',0)))"; }
for round in 1 2; do
  run --workload cfg4 --history 1
  run --workload cfg4
done | tee gpurun_out/r2/ab_ridge_half.txt
