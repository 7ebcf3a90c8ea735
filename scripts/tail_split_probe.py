#!/usr/bin/env python3
"""A/B of the round-5 verdict's "Next" 6: does splitting ONE configs[1] batch into K launches with their own work queues
on K internal streams (the later launches' bulk filling the earlier launches' drain tail) shorten the batch?

The split is emulated above the C-ABI — K contexts on K streams, each solving a contiguous 1/K of the batch, all enqueued
back to back, joined by events on the caller's stream — which is exactly what an internal split of
mi355_lbfgs_minimize_batch would enqueue (same kernels, same grids, own queue words), so the measurement decides whether
the library change is worth making.  Results are bit-identical by construction (a problem's solve does not depend on its
neighbours); checked.  Variants alternate on the same box; the figure is the median over the rounds of the time between
two events on the caller's stream.

    python scripts/tail_split_probe.py [--rounds 12] > gpurun_out/r6_ab_tail.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--m", type=int, default=6)
    args = ap.parse_args()
    import torch
    import cppnumericalsolvers_amd as amd

    dev = torch.device("cuda:0")
    B, n, m = args.batch, args.n, args.m
    ctxs = [amd.Context(0) for _ in range(4)]
    solvers = [amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(), context=c, arithmetic="fma") for c in ctxs]
    x0 = solvers[0].fill_x0(B, n, "std", 20260923)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    main_stream = torch.cuda.current_stream(dev)
    obj = amd.Rosenbrock()

    def run(K, order="contiguous"):
        """One batch as K launches; returns (ms between events on the caller's stream, outputs)."""
        if order == "contiguous":
            parts = [x0[i * (B // K):(i + 1) * (B // K)] for i in range(K)]
        else:  # "interleaved": part i takes problems i, i + K, ... (a different mix of long solves per launch)
            parts = [x0[i::K].contiguous() for i in range(K)]
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(main_stream)
        outs = []
        if K == 1:
            outs.append(solvers[0].minimize(obj, parts[0], want_gradient=False))
        else:
            for i in range(K):
                streams[i].wait_event(start)
                with torch.cuda.stream(streams[i]):
                    outs.append(solvers[i].minimize(obj, parts[i], want_gradient=False))
                    done = torch.cuda.Event()
                    done.record(streams[i])
                main_stream.wait_event(done)
        stop.record(main_stream)
        torch.cuda.synchronize()
        return start.elapsed_time(stop), outs

    variants = [("one launch (today)", 1, "contiguous"), ("2 launches, 2 streams", 2, "contiguous"),
                ("4 launches, 4 streams", 4, "contiguous"), ("2 launches, interleaved problems", 2, "interleaved")]
    # bit-identity of the split results
    _, ref = run(1)
    xr, fr = ref[0][0], ref[0][1]
    for name, K, order in variants[1:]:
        _, outs = run(K, order)
        if order == "contiguous":
            x = torch.cat([o[0] for o in outs])
            f = torch.cat([o[1] for o in outs])
        else:
            x, f = torch.empty_like(xr), torch.empty_like(fr)
            for i, o in enumerate(outs):
                x[i::K], f[i::K] = o[0], o[1]
        assert torch.equal(x, xr) and torch.equal(f, fr), name
    times = {name: [] for name, _, _ in variants}
    for _ in range(2):
        for name, K, order in variants:
            run(K, order)
    for r in range(args.rounds):
        for name, K, order in (variants if r % 2 == 0 else variants[::-1]):   # alternate the order too
            ms, _ = run(K, order)
            times[name].append(ms)
    pn = amd.progress_to_numpy(ref[0][3])
    print("tail split A/B — %d x Rosenbrock-%d, m = %d, fused arithmetic, parity stopping; mean %.1f / max %d iterations"
          % (B, n, m, pn["num_iterations"].mean(), pn["num_iterations"].max()))
    print("device: %s; %d alternating rounds; ms between two events on the caller's stream; every split bit-identical to the "
          "one-launch result" % (torch.cuda.get_device_name(0), args.rounds))
    base = float(np.median(times[variants[0][0]]))
    for name, _, _ in variants:
        t = np.array(times[name])
        print("  %-36s median %.3f ms  (min %.3f, max %.3f)  %6.2f M solves/s  %+5.1f %% vs one launch"
              % (name, np.median(t), t.min(), t.max(), B / np.median(t) / 1e3, (base / np.median(t) - 1) * 100))
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
