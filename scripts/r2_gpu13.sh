#!/bin/bash
# round 2, GPU call 13: staged refill of a finished segment (start point loaded while the wavefront's other problems
# iterate) — 0 = fetch and wait (before), 1 / 3 = passes the segment sits out.  Parity of the default (1), then A/B.
set -u
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu -k "not torchrun and not lbfgsb" > gpurun_out/r2/pytest_staged.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/r2/pytest_staged.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
scripts/ab_variants.sh "cfg2 cfg2:262144 cfg3" 8
cp gpurun_out/ab_variants.txt gpurun_out/r2/ab_staged_fetch.txt
