#!/usr/bin/env python
"""Turns the raw rocprofv3 output merged back under gpurun_out/prof_<tag>/ into the small,
committed summaries under profiles/:
  profiles/<tag>_kernel_stats.txt     rocprofv3 --kernel-trace --stats (per-kernel durations)
  profiles/<tag>_pmc.txt              PMC passes (HBM bytes with the gfx950 corrections, SQ counters)
  profiles/hbm_traffic.json           HBM bytes per launch of the solve kernel (read by bench.py)
usage: python scripts/summarize_profiles.py <tag>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_short(name):
    name = name.replace("void ", "").replace("mi355::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0] if ("lbfgs" in name or "fill_x0" in name or "ridge_" in name) else name[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    lines = ["rocprofv3 --kernel-trace --stats, `python bench.py --steps 2 --warmup 1 --workload <wl>` "
             "(3 solve launches per run; names with a suffix = the same workload with another kernel: _mfma = --ridge-mfma (cfg4 alone is the normal-equation form), "
             "_exact = --arithmetic exact); MI355X, ROCm 7.2", ""]
    traffic = {}
    names = sorted(d[len("stats_"):] for d in os.listdir(src) if d.startswith("stats_") and os.path.isdir(os.path.join(src, d)))
    for wl in names:
        f = os.path.join(src, "stats_%s" % wl, "%s_kernel_stats.csv" % wl)
        if not os.path.exists(f):
            continue
        lines.append("== workload %s" % wl)
        lines.append("%-64s %6s %14s %12s %7s" % ("kernel", "calls", "avg_ms", "total_ms", "%"))
        for r in csv.DictReader(open(f)):
            lines.append("%-64s %6s %14.4f %12.3f %7s" % (kernel_short(r["Name"]), r["Calls"],
                                                          float(r["AverageNs"]) / 1e6,
                                                          float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        lines.append("")
    open(os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag), "w").write("\n".join(lines) + "\n")

    out = ["PMC passes (rocprofv3 --pmc <group> --kernel-trace; one group per run), solve kernel only, "
           "mean over the launches of the run.",
           "HBM bytes: FETCH_SIZE / WRITE_SIZE are in KiB-units of the L2<->fabric request counters.  Per",
           "MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports HALF of the bytes of a coalesced",
           "streaming read, so reads are doubled; WRITE_SIZE is taken as is.  Check against the known byte",
           "count of this kernel: it reads x0 (B*n*8) once and writes x, g (2*B*n*8), f (B*8), progress (B*40).",
           ""]
    # (problems, n, extra read bytes per problem: the ridge right-hand side y_b)
    shapes = {"cfg2": (65536, 32, 0), "cfg3": (131072, 64, 0), "cfg3full": (1048576, 64, 0), "cfg4": (262144, 64, 128 * 8),
              "cfg5": (262144, 32, 0),
              # ridge 1000 x 200 (shared matrix: the solve kernel reads the pre-pass rows, see below) and the own-matrix
              # workload (the solve streams G_b per evaluation: "expected" is only the compulsory part)
              "cfg4big": (32768, 200, 1000 * 8), "cfg4own": (65536, 64, (128 * 64 + 128) * 8),
              # (the workgroup kernel keeps its state in memory: the "expected" figure below is only the compulsory
              #  x0-in / results-out part; what it really moves is the point of this row)
              "wide": (2048, 4096, 0),
              # the SURVEY 8(f) rows of round 6
              "f_hz": (65536, 32, 0), "f_bfgs": (65536, 32, 0), "f_second": (65536, 64, 128 * 8)}
    for wl in names:
        vals = {}
        for grp in ("fetch", "write", "sq", "sq2"):
            f = os.path.join(src, "pmc_%s_%s" % (grp, wl), "%s_counter_collection.csv" % wl)
            if not os.path.exists(f):
                continue
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in ("lbfgs_solve", "lbfgsb_solve", "ridge_mfma_solve", "lbfgsb_fast", "lbfgs_wide")):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    vals["kernel"] = kernel_short(r["Kernel_Name"])
                    vals["vgpr"] = r["VGPR_Count"]
                    vals["lds"] = r["LDS_Block_Size"]
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
        if "FETCH_SIZE" not in vals:
            continue
        key = next((k for k in (wl, "_".join(wl.split("_")[:2]), wl.split("_")[0]) if k in shapes), None)
        if key is None:
            continue
        B, n, extra = shapes[key]
        if vals.get("kernel", "").find("RidgeGram") >= 0:   # the solve kernel reads the pre-pass rows (c_b padded to P, y.y, pad) instead of y_b
            extra = (66 if n <= 64 else 258) * 8
        rd = vals["FETCH_SIZE"] * 1024 * 2.0          # gfx950 correction: x2 on coalesced reads
        wr = vals["WRITE_SIZE"] * 1024
        expect_rd = B * n * 8 + B * extra
        expect_wr = 2 * B * n * 8 + B * 8 + B * 40
        out.append("== workload %s  kernel %s" % (wl, vals.get("kernel")))
        out.append("  FETCH_SIZE %.1f KiB -> corrected read bytes %.3e  (expected x0 read %.3e, ratio %.3f)"
                   % (vals["FETCH_SIZE"], rd, expect_rd, rd / expect_rd))
        out.append("  WRITE_SIZE %.1f KiB -> write bytes %.3e          (expected x,g,f,progress %.3e, ratio %.3f)"
                   % (vals["WRITE_SIZE"], wr, expect_wr, wr / expect_wr))
        out.append("  HBM traffic per launch = %.3e bytes" % (rd + wr))
        for k in sorted(vals):
            if k.startswith("SQ_") or k.startswith("GRBM"):
                out.append("  %-24s %.5g" % (k, vals[k]))
        if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
            busy = vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (vals["GRBM_GUI_ACTIVE"] / 8)
            out.append("  VALU-busy fraction = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs) / (GRBM_GUI_ACTIVE/8 XCDs) = %.3f" % busy)
            out.append("  VALU instructions per wavefront = %.0f; cycles per VALU instruction = %.2f"
                       % (vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], vals["SQ_ACTIVE_INST_VALU"] * 4 / vals["SQ_INSTS_VALU"]))
        # executed lane-flops per launch (solve kernel + the ridge pre-pass) over the rocprof average duration of the
        # solve kernel from the --stats pass of the same command
        ff = os.path.join(src, "pmc_flops_%s" % wl, "%s_counter_collection.csv" % wl)
        if os.path.exists(ff):
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(ff)):
                if any(k in r["Kernel_Name"] for k in ("lbfgs_solve", "lbfgsb_solve", "ridge_mfma_solve", "lbfgsb_fast",
                                                       "lbfgs_wide", "ridge_gram_prepass", "ridge_gram_matrix")):
                    per[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
            c = {k: sum(sum(v) / len(v) for v in kk.values()) for k, kk in per.items()}
            flops = 64.0 * (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) +
                            c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) + \
                512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
            for k in sorted(c):
                out.append("  %-28s %.5g" % (k, c[k]))
            avg_ms = None
            fs = os.path.join(src, "stats_%s" % wl, "%s_kernel_stats.csv" % wl)
            if os.path.exists(fs):
                for r in csv.DictReader(open(fs)):
                    if any(k in r["Name"] for k in ("lbfgs_solve", "lbfgsb_solve", "ridge_mfma_solve", "lbfgsb_fast", "lbfgs_wide")):
                        avg_ms = float(r["AverageNs"]) / 1e6
            out.append("  executed lane-flops per launch = 64 (2 FMA + ADD + MUL + TRANS) + 512 MFMA_MOPS = %.4e" % flops)
            if avg_ms:
                out.append("  frac_executed = %.4e / %.3f ms / 78.6 TFLOP/s = %.3f" % (flops, avg_ms, flops / (avg_ms * 1e-3) / 78.6e12))
        out.append("")
        traffic[wl] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                       "source": "profiles/%s_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                 "FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 note)" % tag,
                       "kernel": vals.get("kernel")}
    open(os.path.join(ROOT, "profiles", "%s_pmc.txt" % tag), "w").write("\n".join(out) + "\n")
    if traffic:
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        merged = {}
        if os.path.exists(path):     # a round that profiles a subset of the workloads keeps the other rows
            try:
                merged = json.load(open(path))
            except ValueError:
                merged = {}
        merged.update(traffic)
        json.dump(merged, open(path, "w"), indent=1)
    print("\n".join(lines[:14]))
    print("\n".join(out))


if __name__ == "__main__":
    main()
