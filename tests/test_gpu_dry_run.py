"""The shared-device dry run of the real bench runtime (round-5 verdict, "Next" 2): `MI355_BENCH_SHARED_DEVICE_DRY_RUN=N
python bench.py` starts N ranks under torch.distributed.run that ALL use device 0, over gloo — the real GpuRuntime, real
kernels, the 3-word all-reduce per step, the rank-0-only rocprofv3 counter passes and CPU legs with the other ranks parked,
the strong configs[2] row, the line assembly.  Everything a first 8-GPU contact would exercise except RCCL itself and the
other seven devices."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(ranks, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MI355_BENCH_SHARED_DEVICE_DRY_RUN"] = str(ranks)
    env["MI355_BENCH_CPU_BUDGET_S"] = "1.0"
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + list(extra),
                       env=env, capture_output=True, text=True, timeout=600)
    return r, time.perf_counter() - t0


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 8])
def test_shared_device_dry_run_of_the_real_bench_runtime(ranks):
    r, seconds = _run(ranks)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line, the others none
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 1               # never a multi-GPU line
    mg = d["multi_gpu"]
    assert mg["ranks_in_this_run"] == ranks and mg["rccl_ranks"] == ranks and mg["collective_backend"] == "gloo"
    assert mg["measured"] is False and len(set(mg["devices"])) == 1 and len(mg["devices"]) == ranks
    assert mg["problems_per_rank"] == [65536] * ranks              # weak scaling: every rank its own shard
    assert d["config"]["problems_total"] == 65536 * ranks and d["config"]["all_converged"]
    # the strong configs[2] row ran over all ranks (contiguous shards of the 1,048,576 problems)
    row = d["secondary_cfg3full_strong"]
    assert row["dry_run"] and row["ranks"] == ranks and row["n_gpus"] == 1 and row["rccl_ranks"] == ranks
    assert sum(row["problems_per_rank"]) == 1048576 and row["global_record"]["total"] == 1048576
    assert row["global_record"]["unconverged"] == 0
    assert d["north_star"]["n_gpus"] == 1
    # rank 0's counter passes ran under the launcher while the others were parked, and so did the CPU legs — on the
    # box's cores, not on the launcher's OMP_NUM_THREADS=1
    rf = d["roofline"]
    assert rf["traffic"] is not None and "measured in this run" in rf["traffic_source"], rf
    assert d["roofline_valu"]["valu_busy"] is not None and d["roofline_valu"]["executed_flops"] is not None
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= min(8, os.cpu_count() or 1)
    assert d["config"]["parity_vs_cpu_sample"]["max_abs_dx"] <= 1e-6
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r6_dry_run_n%d.json" % ranks), "w") as fh:
            fh.write(lines[0] + "\n")
    print("\nshared-device dry run, %d ranks: %.1f s wall; value %.3g solves/s (ONE GPU shared by %d processes), strong row "
          "%.3g solves/s" % (ranks, seconds, d["value"], ranks, row["value"]))


@pytest.mark.gpu
def test_dry_run_is_refused_under_gpus_n():
    r, _ = _run(2, ["--gpus", "2"])
    assert r.returncode != 0 and "never a --gpus N line" in (r.stdout + r.stderr)
