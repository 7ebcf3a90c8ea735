#!/bin/bash
# A/B on ONE GPU box: the fused augmented-Lagrangian kernels with the y half of the inner solver's history in registers
# (MR = 10: 64-72 B of scratch per lane after the cold-lane fix) against both halves in the LDS ring
# (-DMI355_AL_FUSED_LDS_RING: 0 B).  Libraries built in the authoring container:
#   cppnumericalsolvers_amd/variants/lib_base.so, lib_al_lds_ring.so  (scripts/ab_variants.sh has the recipe)
# usage (GPU box): scripts/ab_al_ring.sh
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r5_ab_al_ring.txt
: > $OUT
for round in 1 2; do
for lib in base al_lds_ring; do
  for n in 12 30 64 100; do
    MI355_LBFGS_LIBRARY=$PWD/cppnumericalsolvers_amd/variants/lib_$lib.so python scripts/auglag_bench.py --n $n --batch 16384 \
        --steps 3 --cpu-sample 64 --loop fused 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s n=%-4d %10.0f solves/s %9.3f ms  inner its %.1f  dx %.2g' % ('$lib', $n, d['value'], d['ms_per_step'], d['inner_iterations_mean'], d['parity']['max_abs_dx_vs_oracle_sequential']))" >> $OUT
  done
done
done
cat $OUT
