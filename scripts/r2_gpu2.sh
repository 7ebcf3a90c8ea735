#!/bin/bash
# round 2, GPU call 2: the new test files, then the default bench line (live counters + CPU legs), then the whole GPU suite
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_fma.py tests/test_gpu_reference_and_dist.py -x -q -m gpu > gpurun_out/r2/pytest_new.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r2/pytest_new.log
( time timeout 900 python bench.py ) > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r2/bench_default.err
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu.log 2>&1
echo "suite rc=$?"; tail -5 gpurun_out/r2/pytest_gpu.log
