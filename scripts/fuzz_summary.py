#!/usr/bin/env python3
"""Summarise the JSON lines of scripts/fuzz_parity.py / fuzz_auglag.py runs into profiles/<tag>_fuzz_parity.txt.

    python scripts/fuzz_summary.py r6          # gpurun_out/r6_fuzz_*.jsonl -> profiles/r6_fuzz_parity.txt
    python scripts/fuzz_summary.py             # round 5's naming: gpurun_out/fuzz_*.jsonl -> profiles/r5_fuzz_parity.txt"""
import collections
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
    pattern = "fuzz_*.jsonl" if tag == "r5" else tag + "_fuzz_*.jsonl"
    out = []
    out.append(tag + ": randomised differential campaigns, device solve vs the CPU twin of the same summation tree, compared for EQUALITY")
    out.append("(scripts/fuzz_parity.py, scripts/fuzz_auglag.py on 1 x MI355X; a short draw of both runs in every `pytest -m gpu`: tests/test_gpu_fuzz.py).")
    out.append("fuzz_parity draws per trial: solver (Lbfgs / Lbfgsb reference-order / Lbfgsb relaxed / Bfgs / Lbfgs Second mode from the functor with the")
    out.append("condition_hessian test / the normal-equation and matrix-core ridge forms / the n > 256 workgroup kernel), objective (Rosenbrock-N / diagonal")
    out.append("quadratic / shared-matrix ridge), n (edges of every mapping over-weighted), m, line search, arithmetic policy, explicit lanes x coordinates")
    out.append("mapping, history placement, every stopping field, batch 1..300, start points, per-coordinate boxes with infinite and pinned entries;")
    out.append("compared: x, f, g, status, num_iterations, nfev (+ sum_k) of every problem.  fuzz_auglag draws random term tables (every primitive kind,")
    out.append("sums, forms), constraint families up to the mapping's capacity, configuration, initial multipliers, device loop, line search; compared: x,")
    out.append("lambda, mu, penalty, max_violation, KKT norm and every field of the progress record.")
    out.append("A refusal (a shape the library has no kernel for: MI355_ERR_UNSUPPORTED / INVALID_ARGUMENT) is counted, never compared.\n")
    total_trials = total_problems = total_mismatch = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern))):
        refusals = collections.Counter()
        per = collections.Counter()
        problems = 0
        longest = []
        summary = None
        for line in open(path):
            r = json.loads(line)
            if "summary" in r:
                summary = r["summary"]
            elif "refused" in r:
                refusals[(r.get("solver", r.get("kind")), r["refused"].split(": ", 2)[-1][:96])] += 1
            else:
                per[r.get("solver", r.get("kind"))] += 1
                problems += r["B"]
                longest.append(r.get("iterations_max", r.get("inner_max", 0)))
        if summary is None:
            continue
        total_trials += summary["compared"]
        total_problems += problems
        total_mismatch += summary["mismatch"]
        out.append("%s  seed %d: %d trials compared (%d problems), %d MISMATCHES, %d refused, %.0f s" % (
            os.path.basename(path), summary["seed"], summary["compared"], problems, summary["mismatch"], summary["refused"], summary["seconds"]))
        out.append("   compared: " + ", ".join("%s %d" % kv for kv in sorted(per.items())))
        out.append("   iterations of the longest solve of a trial: median %d, 99%% %d, max %d" % tuple(np.percentile(longest, [50, 99, 100])))
        for (who, why), count in refusals.most_common():
            out.append("   refused %4d  %-14s %s" % (count, who, why))
        out.append("")
    out.append("TOTAL: %d trials, %d problems, %d mismatches" % (total_trials, total_problems, total_mismatch))
    if total_trials == 0:
        print("no runs found under gpurun_out/" + pattern)
        return 1
    open(os.path.join(ROOT, "profiles", tag + "_fuzz_parity.txt"), "w").write("\n".join(out) + "\n")
    print(out[-1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
