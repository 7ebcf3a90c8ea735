#!/bin/bash
# round 2, GPU call 11: More-Thuente cstep with operands selected first (one arithmetic path) vs the reference's
# branch chain: parity of the new build, then an A/B on every workload that runs the search.
set -u
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu -k "not torchrun" > gpurun_out/r2/pytest_cstep.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/r2/pytest_cstep.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
scripts/ab_variants.sh "cfg2 cfg3 cfg4 cfg5" 6
cp gpurun_out/ab_variants.txt gpurun_out/r2/ab_cstep.txt
