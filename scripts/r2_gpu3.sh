#!/bin/bash
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_cpp_host_api.py -x -q -m gpu > gpurun_out/r2/pytest_boundary.log 2>&1
echo "boundary tests rc=$?"; tail -25 gpurun_out/r2/pytest_boundary.log
timeout 600 python bench.py --no-counters > gpurun_out/r2/bench_pcie.json 2> gpurun_out/r2/bench_pcie.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_pcie.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['pcie_inclusive_host_entry'])
PY
