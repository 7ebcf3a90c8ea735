import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def reference():
    """tests/ref_lib.py: the reference binary (oracle/_ref/libref.so, built where the reference tree is; travels to the GPU
    box with the snapshot)."""
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so did not travel to this box")
    return ref_lib


@pytest.fixture(scope="session")
def gpu_solver_factory():
    """Factory of BatchedLbfgs objects sharing one context on cuda:0."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import cppnumericalsolvers_amd as amd
    ctx = amd.Context(0)

    def make(**kw):
        # the bit-parity tests pin the exact arithmetic; tests of the fused kernels ask for arithmetic="fma"
        kw.setdefault("arithmetic", "exact")
        return amd.BatchedLbfgs(context=ctx, **kw)

    yield make
    ctx.close()
