"""The cppoptlib-shaped C++17 host API (include/cppoptlib/...): compiled with plain g++
against the C-ABI library everywhere; executed on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _stamp():
    """Content hash of everything the C++ test binaries are built from (mtimes do not survive the copy to the GPU box,
    contents do — as for the library itself, cppnumericalsolvers_amd/_build.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "cppoptlib", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "cppoptlib", "*", "*.h")) + glob.glob(os.path.join(CPP, "*.cc")) +
                   glob.glob(os.path.join(CPP, "*.h")) + [os.path.join(CPP, "Makefile")] +
                   glob.glob(os.path.join(ROOT, "oracle", "eigen_shim", "Eigen", "*")) +   # (the Eigen-branch builds)
                   glob.glob(os.path.join(ROOT, "cppnumericalsolvers_amd", "*.so.srchash")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _make(target):
    """`make all` / `make run` of tests/cpp — skipping the twenty-one g++ compilations (80 s of GPU-box time, round 4) when
    the binaries that travelled with the tree were built from exactly these sources (`make run-only` then just runs)."""
    stamp_file = os.path.join(CPP, "_build", ".stamp")
    libs = [os.path.join(ROOT, "cppnumericalsolvers_amd", n) for n in ("libmi355_lbfgs.so", "libmi355_lbfgs_hs.so")]
    stamp = _stamp() if all(os.path.exists(l) for l in libs) else None
    fresh = stamp is not None and os.path.exists(stamp_file) and open(stamp_file).read().strip() == stamp
    if fresh and target in ("all", "run"):
        # (the libraries the binaries link against are the ones their stamp was taken with: __graft_entry__.build() — which
        #  also re-makes the oracle, a minute of g++ on a box whose copy of the tree has fresh mtimes — has nothing to do)
        probe = subprocess.run(["make", "-s", "-C", CPP, "have-all"], capture_output=True, text=True)
        if probe.returncode == 0:
            if target == "all":
                return probe
            return subprocess.run(["make", "-s", "-C", CPP, "run-only"], capture_output=True, text=True)
    import __graft_entry__ as ge
    ge.build()
    stamp = _stamp()
    r = subprocess.run(["make", "-s", "-j8", "-C", CPP, "all"], capture_output=True, text=True)
    if r.returncode == 0:
        with open(stamp_file, "w") as fh:
            fh.write(stamp)
        if target == "run":
            r = subprocess.run(["make", "-s", "-C", CPP, "run-only"], capture_output=True, text=True)
    return r


def test_host_headers_compile_and_link_with_gxx():
    r = _make("all")
    assert r.returncode == 0, r.stdout + r.stderr
    for t in ("quickstart_test", "verify_lbfgs_test", "verify_lbfgsb_test", "cstep_test",
              "readme_ridge_test", "hager_zhang_test", "verify_bfgs_test", "augmented_lagrangian_test",
              "quickstart_test_noexcept", "device_path_test", "batch_functions_test", "host_glue_stress_test",
              "shared_params_check_test", "sweep_env_test", "function_expr_test", "penalty_expressions_test",
              # the CPPOPTLIB_MI355_HAVE_EIGEN branch of the drop-in headers (an <Eigen/Core> on the include path)
              "eigen/quickstart_test", "eigen/readme_ridge_test", "eigen/function_expr_test",
              "eigen/augmented_lagrangian_test", "eigen/penalty_expressions_test"):
        assert os.path.exists(os.path.join(CPP, "_build", t))


@pytest.mark.gpu
def test_host_api_cpp_tests_run_on_gpu():
    r = _make("run")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ALL PASSED") == 23, r.stdout   # 18 binaries + the five Eigen-branch builds
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("   (")))   # seconds per binary (pytest -s / -rP)


def test_batches_that_do_not_share_their_parameters_are_refused_on_the_host():
    """tests/cpp/shared_params_check_test.cc: every refusal of MinimizeBatch(functions, states) — a sweep over lambda, one
    function differing anywhere in its blob at an unsampled position, MI355_ARITH_EXACT with own matrices, an
    ill-conditioned own-matrix batch — is decided before the device is touched, so the binary runs here too."""
    r = _make("all")
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(CPP, "_build", "shared_params_check_test")], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


def test_penalty_helpers_and_clipped_expression_nodes_on_the_host():
    """tests/cpp/penalty_expressions_test.cc: QuadraticEqualityPenalty / InequalityPenaltyGe / Lt, Form*Part, ToPenalty,
    ConstExpression, MinZeroExpression / MaxZeroExpression, f - g (reference function_penalty.h:40-61, 97-222,
    function_expressions.h:46-87, 148-196, 319-399) against closed forms and against the composite ToAugmentedLagrangian
    returns — host-side nodes, so both header branches run here."""
    r = _make("all")
    assert r.returncode == 0, r.stdout + r.stderr
    for binary in ("penalty_expressions_test", os.path.join("eigen", "penalty_expressions_test")):
        r = subprocess.run([os.path.join(CPP, "_build", binary)], capture_output=True, text=True)
        assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


def test_cppopt_sweep_build_reads_the_default_preset_from_the_environment():
    """tests/cpp/sweep_env_test.cc (-DCPPOPT_SWEEP, the reference's progress.h:359-381): host only, runs here."""
    r = _make("all")
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(CPP, "_build", "sweep_env_test")], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr


def _compiles(source, tmp_path):
    src = tmp_path / "probe.cc"
    src.write_text(source)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    return r.returncode == 0, r.stderr


USER_FUNCTOR = """
#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgs.h"
using namespace cppoptlib::function;
// an arbitrary host functor, as the reference accepts them: no device twin
class Mine : public FunctionXd<Mine> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* g = nullptr) const {
    if (g) *g = x;
    return x[0];
  }
};
"""


def test_functions_without_a_device_twin_are_rejected_at_compile_time(tmp_path):
    """No CPU fallback: a functor without a device twin cannot be handed to the solvers or used as a term."""
    ok, _ = _compiles(USER_FUNCTOR + "int main() { SquaredNorm<> c; ConstrainedOptimizationProblem<> p(c, {c - 1.0}); "
                      "cppoptlib::solver::Lbfgs<Rosenbrock<>> s; (void)p; (void)s; return 0; }", tmp_path)
    assert ok                                                         # the probe itself is sound
    ok, err = _compiles(USER_FUNCTOR + "int main() { cppoptlib::solver::Lbfgs<Mine> s; (void)s; return 0; }", tmp_path)
    assert not ok and "no device twin" in err
    # Since round 6 the problem stores the reference's type-erased FunctionExpr<TScalar, Mode, TDim>: ANY function
    # converts into it (as in the reference, function_base.h:210-232) and evaluates on the host, so `m - 1.0` and a
    # right-nested sum compile; AugmentedLagrangian refuses them at RUN time with the reason
    # (tests/cpp/function_expr_test.cc "no twin", on the GPU box).  A mode UPGRADE stays a compile error.
    ok, err = _compiles(USER_FUNCTOR + "int main() { Mine m; SquaredNorm<> c; "
                        "ConstrainedOptimizationProblem<> p(c, {m - 1.0}); (void)p; return 0; }", tmp_path)
    assert ok, err
    ok, err = _compiles(USER_FUNCTOR + "int main() { SquaredNorm<> c; LinearForm<> l(std::vector<double>{1.0}); "
                        "ConstrainedOptimizationProblem<> p(c + (l + c)); (void)p; return 0; }", tmp_path)
    assert ok, err
    ok, err = _compiles(USER_FUNCTOR + "int main() { Mine m; FunctionExpr<double, DifferentiabilityMode::Second> e(m); "
                        "(void)e; return 0; }", tmp_path)
    assert not ok and "Differentiability mode mismatch" in err
