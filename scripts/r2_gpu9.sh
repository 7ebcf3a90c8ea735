#!/bin/bash
set -u
mkdir -p gpurun_out/r2
bash scripts/ab_variants.sh "cfg2 cfg2:262144 cfg3 cfg4 cfg5" 5
mv gpurun_out/ab_variants.txt gpurun_out/r2/ab_sgpr.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu3.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r2/pytest_gpu3.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
