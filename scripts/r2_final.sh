#!/bin/bash
# final round-2 measurements with the committed kernels: the default bench line (as the driver runs it), the other
# workloads, rocprofv3 kernel stats + PMC passes
set -u
mkdir -p gpurun_out/r2f
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2f/bench_cfg2.json 2> gpurun_out/r2f/bench_cfg2.err; echo "cfg2 rc=$?"
for wl in cfg3 cfg3full cfg4 cfg5; do
  timeout 900 python bench.py --workload $wl --no-secondary > gpurun_out/r2f/bench_$wl.json 2> gpurun_out/r2f/bench_$wl.err; echo "bench $wl rc=$?"
done
timeout 600 python bench.py --arithmetic exact --no-secondary > gpurun_out/r2f/bench_cfg2_exact.json 2> gpurun_out/r2f/bench_cfg2_exact.err
timeout 600 python scripts/host_path_probe.py > gpurun_out/r2f/host_path_probe.txt 2>&1
bash scripts/profile_gpu.sh r2 > gpurun_out/r2f/profile.log 2>&1
tail -2 gpurun_out/r2f/profile.log; grep real gpurun_out/r2f/bench_cfg2.err
