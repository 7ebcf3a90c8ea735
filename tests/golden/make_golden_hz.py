#!/usr/bin/env python
"""Generates tests/golden/hager_zhang_reference_vectors.npz by RUNNING THE REFERENCE's
linesearch/hager_zhang.h (oracle/_ref/libref.so = the unmodified /root/reference headers over
oracle/eigen_shim): stand-alone HagerZhang::Search calls on seeded inputs that reach every stage of
hzls (immediate acceptance, bracket expansion, bisection, secant steps, collapsed intervals,
non-finite trial points, the 50-iteration limit, non-descent directions), and
Lbfgs<F, m, HagerZhang>::Minimize runs.  Run in the authoring container; the .npz is committed.
Usage:  python tests/golden/make_golden_hz.py
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

SOLVES = [
    # name, objective, n, m, B, x0 kind, stopping
    ("hz_rosen2_m10_default", "rosenbrock", 2, 10, None, None, "default"),
    ("hz_rosen32_m6_parity", "rosenbrock", 32, 6, 16, "std", "parity"),
    ("hz_rosen64_m10_default", "rosenbrock", 64, 10, 12, "u2", "default"),
    ("hz_quad20_m5_default", "diag_quadratic", 20, 5, 12, "u2", "default"),
]


def search_inputs(n, B, seed):
    """Points, directions and initial steps spread over many orders of magnitude."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-2, 2, (B, n))
    G = np.empty_like(x)
    for b in range(B):
        _, G[b] = oracle_lib.evaluate("rosenbrock", x[b])
    scale = 10.0 ** rng.uniform(-9, 5, (B, 1))
    s = -G * scale + rng.normal(0, 1, (B, n)) * np.abs(G).max(1, keepdims=True) * scale * rng.choice(
        [0, 0.3, 2.0], (B, 1))
    a0 = 10.0 ** rng.uniform(-12, 80, B) * rng.choice([1, 1, 1, 0, -1], B)
    a0[: B // 12] = 10.0 ** rng.uniform(60, 300, B // 12)
    a0[B // 12: B // 4] = 10.0 ** rng.uniform(-3, 1, B // 4 - B // 12)   # the range Lbfgs actually uses
    return x, s, a0


def main():
    out = {}
    for n, B, seed in ((2, 160, 101), (8, 160, 102), (32, 160, 103)):
        x, s, a0 = search_inputs(n, B, seed)
        xo, fo, go, ao = ref_lib.hz_search("rosenbrock", x, s, a0)
        key = "search_n%d" % n
        out[key + ".x"], out[key + ".s"], out[key + ".alpha_init"] = x, s, a0
        out[key + ".x_out"], out[key + ".f_out"], out[key + ".g_out"], out[key + ".alpha_out"] = xo, fo, go, ao
        _, _, _, _, nf = oracle_lib.hz_search("rosenbrock", x, s, a0)
        print("%-12s B=%d  accepted %d  failed %d  evaluations: median %d max %d" % (
            key, B, int((ao > 0).sum()), int((ao <= 0).sum()), int(np.median(nf)), int(nf.max())))
    for name, obj, n, m, B, kind, preset in SOLVES:
        if B is None:
            x0 = np.array([[15.0, 8.0], [-1.0, 2.0], [-1.2, 1.0]])
        else:
            x0 = synthetic_x0_host(B, n, kind, seed=20260923, first_problem=5000)
        params = np.concatenate([np.linspace(1.0, 50.0, n), [5.0]]) if obj == "diag_quadratic" else None
        stop = oracle_lib.parity_stop() if preset == "parity" else ref_lib.default_stop(preset)
        x, f, g, p = ref_lib.minimize_batch(obj, x0, m=m, stop=stop, params=params, linesearch="hager_zhang")
        out[name + ".x0"], out[name + ".x"], out[name + ".f"], out[name + ".g"] = x0, x, f, g
        out[name + ".status"], out[name + ".num_iterations"], out[name + ".nfev"] = (
            p["status"], p["num_iterations"], p["nfev"])
        if params is not None:
            out[name + ".params"] = params
        print("%-26s B=%3d  iters %s  nfev %s" % (name, x0.shape[0], p["num_iterations"][:4], p["nfev"][:4]))
    path = os.path.join(HERE, "hager_zhang_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


SOLVE_TABLE = {c[0]: c for c in SOLVES}

if __name__ == "__main__":
    main()
