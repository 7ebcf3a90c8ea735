#!/bin/bash
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_user_objective.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r2/pytest_user.log 2>&1
echo "user objective + boundary tests rc=$?"; tail -25 gpurun_out/r2/pytest_user.log
timeout 600 python bench.py --no-counters --no-cpu-baseline > gpurun_out/r2/bench_pcie2.json 2> gpurun_out/r2/bench_pcie2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_pcie2.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pcie_inclusive_host_entry'])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu2.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/r2/pytest_gpu2.log
