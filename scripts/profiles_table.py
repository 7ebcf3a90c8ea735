#!/usr/bin/env python
"""Markdown table of a round's measurements from the committed files: profiles/<tag>_bench_n1_<name>.json (bench lines) and
profiles/<tag>_kernel_stats.txt (rocprofv3 --stats averages of the same commands).  usage: python scripts/profiles_table.py r4"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORDER = ["default", "cfg3", "cfg3full", "cfg4", "cfg4_mfma", "cfg4big", "cfg4own", "cfg5", "cfg5_exact", "wide",
         "f_hz", "f_hz_exact", "f_bfgs", "f_second", "f_al", "f_al65k"]   # (f_*: the SURVEY 8(f) rows, round 6)


def rocprof_averages(tag):
    """workload name -> average ms of the solve kernel under rocprofv3 --stats"""
    path = os.path.join(ROOT, "profiles", "%s_kernel_stats.txt" % tag)
    out, wl = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"== workload (\S+)", line)
        if m:
            wl = m.group(1)
            continue
        if wl and re.search(r"(_solve_kernel|lbfgsb_fast_kernel|lbfgs_wide_kernel|_prepass_kernel)", line):
            parts = line.split()
            try:   # columns: kernel, calls, avg_ms, total_ms, %
                key = "prepass" if "_prepass_kernel" in line else "solve"
                out.setdefault(wl, {}).setdefault(key, float(parts[-3]))
            except (ValueError, IndexError):
                pass
    return out


def fmt(v, spec="%.3g", none="—"):
    return none if v is None else spec % v


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
    prof = rocprof_averages(tag)
    print("| workload | kernel | solves/s | kernel ms (HIP events) | rocprofv3 avg ms | **binding fraction** (`roofline.frac_physical`, bound) | model frac (state streaming / 8 TB/s) | measured HBM bytes per launch (frac of 8 TB/s) | issued lane-flops / 78.6 TF | useful / 78.6 TF | VALU-busy | parity sample max\\|dx\\| | CPU port / reference (solves/s, cores) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name in ORDER:
        path = os.path.join(ROOT, "profiles", "%s_bench_n1_%s.json" % (tag, name))
        if not os.path.exists(path):
            continue
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            print("| %s | unreadable: %s |" % (name, e))
            continue
        r, v, c = d["roofline"], d["roofline_valu"], d["config"]
        key = "cfg2" if name == "default" else name
        pk = prof.get(key) or {}
        avg = pk.get("solve")
        dev = ""
        if avg is not None:
            both = avg + pk.get("prepass", 0.0)    # (the HIP events of the normal-equation forms bracket pre-pass + solve)
            dev = (" + %.2f pre-pass" % pk["prepass"] if "prepass" in pk else "") + " (%+.1f %%)" % (100.0 * (both / r["kernel_ms"] - 1.0))
        par = (c.get("parity_vs_cpu_sample") or {}).get("max_abs_dx")
        cpu, ref = d.get("cpu_baseline") or {}, d.get("cpu_reference") or {}
        traffic = r.get("traffic")
        binding = "%s (%s)" % (fmt(r.get("frac_physical"), "%.3f"), r.get("bound_physical", "—"))
        print("| %s | `%s` | %s | %.2f | %s%s | **%s** | %.3f | %s (%s) | %s | %.3f | %s | %s | %s / %s (%s) |" % (
            c["workload"].split(";")[0][:110], r["kernel"], fmt(d["value"], "%.3e"), r["kernel_ms"], fmt(avg, "%.2f"), dev,
            binding, r["frac"], fmt(traffic, "%.3e"), fmt(r.get("hbm_frac_measured"), "%.4f"), fmt(v.get("frac_executed"), "%.3f"),
            v["frac_of_fma_peak"], fmt(v.get("valu_busy"), "%.2f"), fmt(par, "%.2e"), fmt(cpu.get("value"), "%.3g"),
            fmt(ref.get("value"), "%.3g"), cpu.get("cores", "?")))


if __name__ == "__main__":
    main()
