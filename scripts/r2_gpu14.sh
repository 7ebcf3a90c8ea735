#!/bin/bash
# round 2, GPU call 14: Rosenbrock variant for problems that fill their segment (boundary predicates as compile-time
# constants): whole GPU suite with the new default, then same-box A/B against the previous build.
set -u
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu -k "not torchrun" > gpurun_out/r2/pytest_fullseg.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/r2/pytest_fullseg.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
scripts/ab_variants.sh "cfg2 cfg2:262144 cfg3 cfg3full" 5
cp gpurun_out/ab_variants.txt gpurun_out/r2/ab_fullseg.txt
