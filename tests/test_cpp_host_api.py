"""The cppoptlib-shaped C++17 host API (include/cppoptlib/...): compiled with plain g++
against the C-ABI library everywhere; executed on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _make(target):
    import __graft_entry__ as ge
    ge.build()
    return subprocess.run(["make", "-s", "-C", CPP, target], capture_output=True, text=True)


def test_host_headers_compile_and_link_with_gxx():
    r = _make("all")
    assert r.returncode == 0, r.stdout + r.stderr
    for t in ("quickstart_test", "verify_lbfgs_test", "verify_lbfgsb_test", "cstep_test",
              "readme_ridge_test", "hager_zhang_test", "verify_bfgs_test", "augmented_lagrangian_test",
              "quickstart_test_noexcept"):
        assert os.path.exists(os.path.join(CPP, "_build", t))


@pytest.mark.gpu
def test_host_api_cpp_tests_run_on_gpu():
    r = _make("run")
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ALL PASSED") == 9, r.stdout
