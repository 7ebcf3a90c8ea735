#!/bin/bash
# Runs on the GPU box (through gpurun): one full bench.py line (counters + CPU legs) per workload, written to
# gpurun_out/<tag>_bench_n1_<name>.json — the files copied to profiles/ at the end of a round.
#   usage: bash scripts/bench_all.sh <tag> ["name[:bench-arg[:bench-arg...]] ..."]
set -u
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
SPECS=${2:-"default cfg3 cfg3full cfg4 cfg4_mfma:--ridge-mfma cfg4big cfg4own cfg5 cfg5_exact:--arithmetic:exact wide"}
for SPEC in $SPECS; do
  NAME=${SPEC%%:*}
  WL=${NAME%%_*}
  # the SURVEY 8(f) rows: f_hz, f_bfgs, f_second (+ a suffix after the second "_": f_hz_exact = f_hz with extra arguments)
  [[ "$NAME" == f_* ]] && WL=$(echo "$NAME" | cut -d_ -f1,2)
  EXTRA=""
  [[ "$SPEC" == *:* ]] && EXTRA=$(echo "${SPEC#*:}" | tr ':' ' ')
  ARGS="--workload $WL $EXTRA --no-secondary"
  [[ "$NAME" == "default" ]] && ARGS=""          # the driver's own command line: every secondary figure included
  timeout 900 python $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_n1_$NAME.json 2> $OUT/${TAG}_bench_n1_$NAME.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_n1_$NAME.json").read().strip().splitlines()[-1])
    r, v = d["roofline"], d["roofline_valu"]
    print("%-12s %.4g solves/s  kernel %.3f ms  model %.3f  hbm %s  useful %.3f executed %s busy %s  parity dx %s" % (
        "$NAME", d["value"], r["kernel_ms"], r["frac"], r.get("hbm_frac_measured"), v["frac_of_fma_peak"], v.get("frac_executed"),
        v.get("valu_busy"), (d["config"].get("parity_vs_cpu_sample") or {}).get("max_abs_dx")))
except Exception as e:
    print("$NAME FAILED", e)
PY
done
