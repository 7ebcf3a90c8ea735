#!/bin/bash
# A/B of kernel variants on ONE GPU box: every library under cppnumericalsolvers_amd/variants/ is
# a build of the same sources with different -D flags, made in the authoring container with e.g.
#   python -c "from cppnumericalsolvers_amd import _build as b; b.build(extra_flags=['-DX=1'], output=b.PKG_DIR + '/variants/lib_x.so')"
# bench.py picks it up via MI355_LBFGS_LIBRARY.
# usage (on the GPU box): scripts/ab_variants.sh "cfg2 cfg2:262144 cfg3" [steps]
set -u
WL=${1:-"cfg2 cfg3"}
STEPS=${2:-8}
mkdir -p gpurun_out
for round in 1 2; do
for lib in cppnumericalsolvers_amd/variants/lib_*.so; do
  for w in $WL; do
    name=${w%%:*}; batch=0; [[ "$w" == *:* ]] && batch=${w#*:}
    MI355_LBFGS_LIBRARY=$PWD/$lib python bench.py --workload $name --batch $batch --steps $STEPS --warmup 2 \
        --no-cpu-baseline --no-secondary --no-counters 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-28s %-12s %10.0f solves/s %8.3f ms  grid %s lds %s' % ('$(basename $lib)', '$w', d['value'], d['ms_per_step'], c.get('grid_workgroups'), c.get('lds_bytes_per_workgroup')))"
  done
done
done | tee gpurun_out/ab_variants.txt
