#!/usr/bin/env python
"""Generates tests/golden/auglag_reference_vectors.npz by RUNNING THE REFERENCE's augmented-Lagrangian solver
(solver/augmented_lagrangian.h, function_penalty.h, progress.h over oracle/eigen_shim: oracle/_ref/libref.so built
from the unmodified /root/reference headers by oracle/ref_auglag_capi.cpp) on the problems of tests/auglag_lib.py —
among them the reference's own test problems (src/test/verify.cc:290-312; augmented_lagrangian_test.cc:583-621 and
:1198-1275).  Run in the authoring container; the .npz is committed and travels to the GPU box, where
/root/reference does not exist.
Usage:  python tests/golden/make_golden_auglag.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import auglag_lib as al  # noqa: E402


def cases():
    """name -> (problem, x0, penalty0, config overrides, inner solver, bounds, line search); shared with the tests."""
    rng = np.random.default_rng(20260923)
    p16, lo16, hi16 = al.hs016_problem()
    pb, lob, hib = al.boxed_rosenbrock_problem(6)
    p24, lo24, hi24 = al.hs024_problem()
    rng_user = np.random.default_rng(2026092401)   # (its own stream: the older cases keep their starts)
    return {
        # user functors as terms (MI355_AL_TERM_USER): the reference's non-convex tests, each with its own start first
        "hs024_user_box": (p24, np.vstack([[1.0, 0.5], rng_user.uniform(0.5, 4.0, (5, 2)) * [1.0, 0.3]]), 0.0, {}, "lbfgsb",
                           (lo24, hi24), "more_thuente"),
        "hs029_user": (al.hs029_problem(), np.vstack([[1.0, 1.0], rng_user.uniform(0.5, 3.0, (5, 2))]), 0.0, {}, "lbfgs", None,
                       "more_thuente"),
        # ProdExpression as a node of the term table (MI355_AL_PARTS_PRODUCT): Hs029 over the menu, and products in every
        # position / form
        "hs029_product": (al.hs029_product_problem(), np.vstack([[1.0, 1.0], rng_user.uniform(0.5, 3.0, (5, 2))]), 0.0, {},
                          "lbfgs", None, "more_thuente"),
        "product_terms9": (al.product_terms_problem(9), np.random.default_rng(2026092402).uniform(0.1, 1.0, (6, 9)), 0.0,
                           {"outer_num_iterations": 20}, "lbfgs", None, "more_thuente"),
        "circle": (al.circle_problem(), np.vstack([[2.0, 10.0], rng.uniform(-3, 3, (7, 2))]), 1.0, {}, "lbfgs", None, "more_thuente"),
        "simplex12": (al.quadratic_simplex_problem(12), rng.uniform(-1, 1, (8, 12)), 0.0, {}, "lbfgs", None, "more_thuente"),
        "simplex40_hz": (al.quadratic_simplex_problem(40, seed=3), rng.uniform(-1, 1, (6, 40)), 0.0,
                         {"outer_num_iterations": 25}, "lbfgs", None, "hager_zhang"),
        "quadratic_at_12": (al.quadratic_at_12_problem(), np.array([[1.0, 1.0], [3.0, -2.0]]), 1.0, {}, "lbfgs", None, "more_thuente"),
        "three_part7": (al.three_part_problem(7), rng.uniform(-1, 1, (6, 7)), 0.0, {"outer_num_iterations": 20}, "lbfgs", None,
                        "more_thuente"),
        "hs016_box": (p16, np.array([[-2.0, 1.0], [0.0, 0.0]]), 0.0, {}, "lbfgsb", (lo16, hi16), "more_thuente"),
        "boxed_rosenbrock6": (pb, rng.uniform(-1, 1, (6, 6)), 0.0, {"outer_num_iterations": 25}, "lbfgsb", (lob, hib),
                              "more_thuente"),
    }


def run_reference(case):
    p, x0, pen0, cfg_kw, inner, bounds, ls = case
    cfg = al.default_config(**cfg_kw)
    if inner == "lbfgsb":
        return al.ref_box_minimize(p, x0, lower=bounds[0], upper=bounds[1], penalty0=pen0, config=cfg, linesearch=ls)
    return al.ref_minimize(p, x0, penalty0=pen0, config=cfg, linesearch=ls)


def main():
    out = {}
    for name, case in cases().items():
        r = run_reference(case)
        out[name + "/x0"] = case[1]
        for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient"):
            out[name + "/" + k] = r[k]
        for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm"):
            out[name + "/" + k] = r["progress"][k]
        print("%-20s status %s outer iterations %s" % (name, r["progress"]["status"], r["progress"]["num_iterations"]))
    path = os.path.join(HERE, "auglag_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
