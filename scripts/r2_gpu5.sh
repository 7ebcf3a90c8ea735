#!/bin/bash
set -u
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_lbfgsb_wider.py -x -q -m gpu > gpurun_out/r2/pytest_lbfgsb.log 2>&1
echo "lbfgsb wider rc=$?"; tail -15 gpurun_out/r2/pytest_lbfgsb.log | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
timeout 600 python scripts/host_path_probe.py > gpurun_out/r2/host_path_probe.txt 2>&1; grep -v amdgpu gpurun_out/r2/host_path_probe.txt
for wl in cfg3 cfg3full cfg4 cfg5; do
  timeout 900 python bench.py --workload $wl --no-secondary > gpurun_out/r2/bench_$wl.json 2> gpurun_out/r2/bench_$wl.err
  echo "bench $wl rc=$?"
done
timeout 900 python bench.py > gpurun_out/r2/bench_cfg2.json 2> gpurun_out/r2/bench_cfg2.err; echo "bench cfg2 rc=$?"
timeout 900 python bench.py --arithmetic exact --no-secondary > gpurun_out/r2/bench_cfg2_exact.json 2> gpurun_out/r2/bench_cfg2_exact.err
bash scripts/profile_gpu.sh r2 > gpurun_out/r2/profile.log 2>&1; tail -3 gpurun_out/r2/profile.log
