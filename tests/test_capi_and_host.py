"""CPU tests of the boundary and host logic (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from cppnumericalsolvers_amd import capi
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "mi355_lbfgs.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355_(?:lbfgsb?|bfgs|auglag)_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), "symbol %s declared in include/mi355_lbfgs.h is not exported" % name
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    assert lib.mi355_lbfgs_abi_version() == 9


def test_struct_layouts_match_header():
    from cppnumericalsolvers_amd import capi
    assert C.sizeof(capi.Stop) == 64      # static_assert'ed in csrc/mi355_lbfgs.hip
    assert capi.PROGRESS_DTYPE.itemsize == 40
    assert capi.Desc.stop.offset % 8 == 0
    assert capi.AL_PROGRESS_DTYPE.itemsize == 56 and C.sizeof(capi.AlConfig) == 104   # mi355_al_progress / mi355_al_config
    assert C.sizeof(capi.AlProblem) == 96      # + user_params (ABI 7), constraint families (ABI 8)


def test_default_stop_presets_without_gpu():
    from cppnumericalsolvers_amd import capi
    d = capi.default_stop()
    assert (d.num_iterations, d.x_delta, d.gradient_norm, d.past, d.past_delta) == (10000, 1e-9, 1e-5, 3, 1e-6)
    c = capi.default_stop("conservative")
    assert (c.gradient_norm, c.past, c.past_delta) == (5e-6, 5, 1e-10)
    import oracle_lib
    o = oracle_lib.default_stop()
    for name, _ in d._fields_:
        assert getattr(d, name) == getattr(o, name)


def test_no_cpu_fallback_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import cppnumericalsolvers_amd as amd
    with pytest.raises(amd.capi.EngineError) as e:
        amd.Context(0)
    assert e.value.code == amd.capi.ERR_NO_DEVICE


def test_product_never_references_the_oracle():
    """The shipped package must not import/link/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "cppnumericalsolvers_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_lib" not in src and "lbfgs_oracle" not in src and "liboracle" not in src, fn
    hdr = open(os.path.join(ROOT, "include", "mi355_lbfgs.h")).read()
    assert "oracle" not in hdr.lower()


def test_synthetic_x0_generator_is_deterministic_and_sharding_consistent():
    import cppnumericalsolvers_amd as amd
    a = amd.synthetic_x0_host(64, 32, "std")
    b = np.concatenate([amd.synthetic_x0_host(40, 32, "std"), amd.synthetic_x0_host(24, 32, "std", first_problem=40)])
    np.testing.assert_array_equal(a, b)            # shards of a global batch line up
    assert np.all(np.abs(a[:, 0::2] + 1.2) <= 0.1) and np.all(np.abs(a[:, 1::2] - 1.0) <= 0.1)
    u2 = amd.synthetic_x0_host(64, 32, "u2")
    assert u2.min() >= -2.0 and u2.max() < 2.0
    assert amd.synthetic_x0_host(1, 4, "std")[0, 0] == pytest.approx(-1.20697723, abs=1e-8)


def test_algorithmic_bytes_formula():
    import bench
    # B_iter(n,k) = 8 n (6 + 2k); cfg3 at k = m = 10 -> 13,312 B
    assert bench.algorithmic_bytes(64, 1, 10) == 13312
    assert bench.algorithmic_bytes(32, 1, 6) == 4608
    # T iterations with a filling ring, no rejected pair: 8n[T(6+2m) - m(m+1)]
    n, m, T = 64, 10, 381
    sum_k = sum(min(t, m) for t in range(T))
    assert bench.algorithmic_bytes(n, T, sum_k) == 8 * n * (T * (6 + 2 * m) - m * (m + 1))


def test_shard_ranges_partition_the_batch():
    from cppnumericalsolvers_amd.sharded import shard_range
    for B in (0, 1, 7, 65536, 1048576):
        for G in (1, 2, 3, 4, 8):
            r = [shard_range(B, g, G) for g in range(G)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_composite_objective_refuses_a_problem_with_constraint_families():
    """AugLagComposite describes MI355_OBJ_AL_COMPOSITE by the term table only: a ConstrainedProblem that also carries
    constraint families is refused on the host instead of silently losing them (round-5 advisor finding)."""
    import numpy as np
    import pytest
    from cppnumericalsolvers_amd import AugLagComposite, ConstrainedProblem
    T = ConstrainedProblem.term
    plain = ConstrainedProblem(3, T("rosenbrock"), [T("linear", "value_minus_k", 0.5, a=[1.0, 1.0, 1.0])])
    assert AugLagComposite(plain).params.size > 0
    # (the new least-squares residual primitive is a table term like the others)
    ls = ConstrainedProblem(3, T([("squared_affine", [1.0, 2.0, 0.0], 4.0), ("squared_affine", [3.0, 1.0, 1.0], 5.0)]))
    assert int(ls.kinds[0]) == 4 and AugLagComposite(ls).params.size > 0
    with_family = ConstrainedProblem(3, T("rosenbrock"), [], [], family_inequality=(np.eye(3), np.zeros(3)))
    with pytest.raises(ValueError, match="constraint families"):
        AugLagComposite(with_family)
