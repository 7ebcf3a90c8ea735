#!/usr/bin/env python
"""Generates tests/golden/lbfgs_reference_vectors.npz by RUNNING THE REFERENCE
(oracle/_ref/libref.so = the unmodified /root/reference headers over
oracle/eigen_shim) on seeded inputs.  Run in the authoring container, where
/root/reference exists; the .npz is committed so the vectors travel to the GPU
box.  Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
import ref_lib  # noqa: E402
from cppnumericalsolvers_amd.engine import synthetic_x0_host  # noqa: E402

CASES = [
    # name, objective, n, m, B, x0 kind, stopping preset
    ("rosen2_default", "rosenbrock", 2, 10, None, None, "default"),
    ("quad2_default", "diag_quadratic", 2, 10, None, None, "default"),
    ("rosen8_m4_default", "rosenbrock", 8, 4, 24, "u2", "default"),
    ("rosen32_m6_default", "rosenbrock", 32, 6, 24, "u2", "default"),
    ("rosen32_m6_conservative", "rosenbrock", 32, 6, 16, "std", "conservative"),
    ("rosen32_m6_parity", "rosenbrock", 32, 6, 24, "std", "parity"),
    ("rosen64_m10_parity", "rosenbrock", 64, 10, 16, "std", "parity"),
    ("rosen48_m10_parity_u2", "rosenbrock", 48, 10, 16, "u2", "parity"),
    ("quad20_m5_default", "diag_quadratic", 20, 5, 16, "u2", "default"),
]


def stop_for(preset):
    if preset == "parity":
        return oracle_lib.parity_stop()
    return ref_lib.default_stop(preset)


def main():
    out = {}
    for name, obj, n, m, B, kind, preset in CASES:
        if name == "rosen2_default":
            x0 = np.array([[15.0, 8.0], [-1.0, 2.0], [-1.2, 1.0]])   # src/test/verify.cc:168-173 + classic
        elif name == "quad2_default":
            x0 = np.array([[-10.0, 2.0]])                             # README.md:30-36
        else:
            x0 = synthetic_x0_host(B, n, kind, seed=20260923, first_problem=1000)
        params = None
        if obj == "diag_quadratic":
            params = np.array([5.0, 100.0, 5.0]) if n == 2 else np.concatenate(
                [np.linspace(1.0, 50.0, n), [5.0]])
        x, f, g, p = ref_lib.minimize_batch(obj, x0, m=m, stop=stop_for(preset), params=params)
        out[name + ".x0"] = x0
        out[name + ".x"] = x
        out[name + ".f"] = f
        out[name + ".g"] = g
        out[name + ".status"] = p["status"]
        out[name + ".num_iterations"] = p["num_iterations"]
        out[name + ".nfev"] = p["nfev"]
        if params is not None:
            out[name + ".params"] = params
        print("%-28s B=%3d  iters %s  nfev %s" % (name, x0.shape[0], p["num_iterations"][:4], p["nfev"][:4]))
    # cstep: random inputs through the reference's cstep
    rng = np.random.default_rng(11)
    recs, outs = [], []
    for _ in range(512):
        stx = rng.uniform(0, 2)
        stp = stx + rng.uniform(0.01, 3) * rng.choice([1.0, 1.0, -0.3])
        dx = -np.sign(stp - stx) * rng.uniform(0.01, 5)
        fx = rng.normal()
        fp = fx + rng.normal() * 0.5
        dp = rng.normal() * 3
        brackt = bool(rng.random() < 0.4)
        sty = stp + np.sign(stp - stx) * rng.uniform(0.01, 2) if brackt else 0.0
        fy = fx + abs(rng.normal())
        dy = rng.normal()
        lo, hi = min(stx, sty if brackt else stx), max(stx, sty if brackt else stp * 5) + 1.0
        r = ref_lib.cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, lo, hi)
        recs.append([stx, fx, dx, sty, fy, dy, stp, fp, dp, float(brackt), lo, hi])
        outs.append([r["rc"], r["info"], float(r["brackt"]), r["stx"], r["fx"], r["dx"], r["sty"], r["fy"],
                     r["dy"], r["stp"]])
    out["cstep.in"] = np.array(recs)
    out["cstep.out"] = np.array(outs)
    path = os.path.join(HERE, "lbfgs_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


CASE_TABLE = {c[0]: c for c in CASES}

if __name__ == "__main__":
    main()
