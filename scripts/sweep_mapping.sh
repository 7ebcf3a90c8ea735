#!/bin/bash
# Sweep (workload, lanes W, elems E, history placement H: 1 = LDS ring, 2 = y half in registers).
cd ${GRAFT_REPO_ROOT:-.}
CONFIGS=${SWEEP_CONFIGS:-"cfg2:32:1:1 cfg2:16:2:1 cfg2:16:2:2 cfg2:8:4:1 cfg2:8:4:2 cfg3:64:1:1 cfg3:32:2:1 cfg3:32:2:2 cfg3:16:4:1 cfg3:16:4:2"}
for cfg in $CONFIGS; do
  IFS=: read WL W E H <<< "$cfg"
  echo -n "$WL W=$W E=$E H=$H : "
  timeout 300 python bench.py --workload $WL --lanes $W --elems $E --history $H --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys,json
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); c=r['config']; ok=True
        print('%.0f solves/s  %.2f ms  frac %.3f  grid %d  lds %d  yreg %d' % (r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'], c['grid_workgroups'], c['lds_bytes_per_workgroup'], c['y_columns_in_registers']))
if not ok: print('FAILED')
"
done
