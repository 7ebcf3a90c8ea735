cd $GRAFT_REPO_ROOT
for cfg in "cfg2 32 1" "cfg2 16 2" "cfg2 8 4" "cfg2 64 1" "cfg3 64 1" "cfg3 32 2" "cfg3 16 4"; do
  set -- $cfg
  echo "== $1 W=$2 E=$3"
  timeout 300 python bench.py --workload $1 --lanes $2 --elems $3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'], 'grid', r['config']['grid_wavefronts'], 'lds', r['config']['lds_bytes_per_wavefront'])
"
done
