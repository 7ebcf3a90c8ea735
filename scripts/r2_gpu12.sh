#!/bin/bash
# round 2, GPU call 12: L-BFGS-B subspace step — WZ WZ^T in full 16-sum butterflies with operands zeroed outside the
# free set, and M^-1 (WZ r) riding along with the column solves: parity, then same-box A/B on config 5.
set -u
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu -k "lbfgsb or auglag or boundary" > gpurun_out/r2/pytest_lbfgsb12.log 2>&1
echo "gpu tests rc=$?"; tail -5 gpurun_out/r2/pytest_lbfgsb12.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
scripts/ab_variants.sh "cfg5" 5
cp gpurun_out/ab_variants.txt gpurun_out/r2/ab_lbfgsb12.txt
