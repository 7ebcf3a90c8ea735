#!/bin/bash
# round 2, GPU call 1: config 3 at its full size, A/B of build variants, phase shares
set -u
mkdir -p gpurun_out/r2
python bench.py --workload cfg3 --batch 1048576 --steps 3 --warmup 1 --no-secondary > gpurun_out/r2/cfg3_full.json 2> gpurun_out/r2/cfg3_full.err
bash scripts/ab_variants.sh "cfg2 cfg2:262144 cfg3" 6
mv gpurun_out/ab_variants.txt gpurun_out/r2/
for lib in lphases lphases_fma; do
  for args in "1" "65536" "131072 64 10" "1 64 10"; do
    echo "== $lib $args"
    MI355_LBFGS_LIBRARY=$PWD/cppnumericalsolvers_amd/variants_prof/lib_$lib.so python scripts/lbfgs_phases.py $args
  done
done > gpurun_out/r2/phases.txt 2>&1
tail -3 gpurun_out/r2/cfg3_full.err
