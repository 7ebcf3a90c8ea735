#!/bin/bash
# final round-2 measurements, second pass (after the L-BFGS-B subspace, ridge A^T R and generality changes): the full
# GPU test suite, then scripts/r2_final.sh (bench lines for every workload, rocprofv3 kernel stats + PMC passes)
set -u
mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2f/pytest_gpu_final.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/r2f/pytest_gpu_final.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
bash scripts/r2_final.sh
