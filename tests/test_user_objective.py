"""The user-objective example on the CPU: the oracle twin of the SVM functor (oracle::SvmSquaredHinge) against the
reference's Lbfgs on the functor of src/examples/svm_primal_lbfgs.cc (oracle/_ref), and the build plumbing."""
import os

import numpy as np
import pytest

import oracle_lib as O
import ref_lib
import svm_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_svm_twin_equals_the_reference_example_functor_bit_for_bit():
    if not ref_lib.available() or not hasattr(ref_lib.lib(), "ref_svm_minimize_batch"):
        pytest.skip("oracle/_ref/libref.so without the SVM entry")
    X, y = svm_data.two_blobs()
    p = svm_data.params(X, y, C=1.0)
    n = X.shape[1] + 1
    x0 = np.vstack([np.zeros(n), np.random.default_rng(1).normal(size=(7, n))])   # the example starts at the origin
    for stop in (O.default_stop(), O.parity_stop()):
        xs, fs, gs, ps = O.minimize_batch("svm_squared_hinge", x0, m=10, stop=stop, params=p)
        xr, fr, gr, pr = ref_lib.svm_minimize_batch(p, x0, m=10, stop=stop)
        np.testing.assert_array_equal(xs, xr)
        np.testing.assert_array_equal(fs, fr)
        np.testing.assert_array_equal(gs, gr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(ps[k], pr[k])
    # the classifier separates the blobs (the example reports ~97 % on its data)
    w, b = xs[0, :-1], xs[0, -1]
    assert np.mean(np.sign(X @ w + b) == y) > 0.9
    # the device's summation tree (butterfly over the coordinates) stays within the tolerance
    xb, fb, _, _ = O.minimize_batch("svm_squared_hinge", x0, m=10, stop=O.parity_stop(), params=p, reduction="butterfly",
                                    width=8)
    xs, fs, _, _ = O.minimize_batch("svm_squared_hinge", x0, m=10, stop=O.parity_stop(), params=p)
    assert np.max(np.abs(xb - xs)) <= 1e-6 and np.max(np.abs(fb - fs)) <= 1e-6


def _svm_box(n):
    # weights held in [-0.25, 0.25] (active at the solution), the offset free
    lo = np.concatenate([np.full(n - 1, -0.25), [-1e3]])
    hi = np.concatenate([np.full(n - 1, 0.25), [1e3]])
    return lo, hi


def test_svm_twin_under_lbfgsb_equals_the_reference_bit_for_bit():
    """The same functor under the box-constrained solver: oracle::Lbfgsb (reference sort order) against the reference's
    Lbfgsb<F, 5> on the example's functor."""
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so not built")
    X, y = svm_data.two_blobs()
    p = svm_data.params(X, y, C=1.0)
    n = X.shape[1] + 1
    lo, hi = _svm_box(n)
    x0 = np.vstack([np.zeros(n), np.random.default_rng(2).normal(size=(9, n))])
    for stop in (O.lbfgsb_default_stop(), O.parity_stop()):
        xs, fs, gs, ps = O.lbfgsb_minimize_batch("svm_squared_hinge", x0, m=5, stop=stop, params=p, lower=lo, upper=hi,
                                                 std_sort_order=True)
        try:
            xr, fr, gr, pr = ref_lib.lbfgsb_minimize_batch("svm_squared_hinge", x0, m=5, stop=stop, lower=lo, upper=hi, params=p)
        except ValueError:
            pytest.skip("oracle/_ref/libref.so without the SVM entry of Lbfgsb")
        np.testing.assert_array_equal(xs, xr)
        np.testing.assert_array_equal(fs, fr)
        np.testing.assert_array_equal(gs, gr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(ps[k], pr[k])
        assert np.all(xs <= hi) and np.all(xs >= lo) and np.any(np.abs(xs[:, :-1]) == 0.25)   # bounds are active
        xb, fb, _, _ = O.lbfgsb_minimize_batch("svm_squared_hinge", x0, m=5, stop=stop, params=p, lower=lo, upper=hi,
                                               reduction="butterfly", width=16)
    assert np.max(np.abs(xb - xs)) <= 1e-6 and np.max(np.abs(fb - fs)) <= 1e-6


def test_user_objective_translation_units_are_generated(tmp_path):
    from cppnumericalsolvers_amd import _build
    hdr = os.path.join(ROOT, "examples", "user_objective_svm", "svm_squared_hinge.hpp")
    paths = _build.user_objective_sources([dict(name="svm", header=hdr, type="user_examples::SvmSquaredHinge", id=100)],
                                          str(tmp_path))
    names = [os.path.basename(q) for q in paths]
    # Lbfgsb: one unit per kernel of the default shapes (n <= 64, m <= 5: three reference-order + three relaxed kernels)
    # + the dispatch table; then the four lanes-per-problem units of Lbfgs / Bfgs
    assert len(paths) == 6 + 1 + 4
    assert names[-4:] == ["user_svm_w8.hip", "user_svm_w16.hip", "user_svm_w32.hip", "user_svm_w64.hip"]
    assert "user_svm_lbfgsb_exact_w16_e4_m5.hip" in names and "user_svm_lbfgsb_relaxed_w16_e1_m5.hip" in names
    table = open(paths[names.index("user_svm_lbfgsb.hip")]).read()
    assert "UserLbfgsbRegistration" in table and "user_100_lbfgsb_exact_w16_e2_m5(ctx, args, stream)" in table
    k = open(paths[names.index("user_svm_lbfgsb_exact_w16_e4_m5.hip")]).read()
    assert "launch_lbfgsb<4, user_examples::SvmSquaredHinge, 5, MI355_LS_MORE_THUENTE, NoOuterLoop, 16>" in k and hdr in k
    k = open(paths[names.index("user_svm_lbfgsb_relaxed_w16_e2_m5.hip")]).read()
    assert "launch_lbfgsb_fast_user<2, user_examples::SvmSquaredHinge, 5>" in k and "lbfgsb_fast_dispatch.hpp" in k
    wide = _build.user_objective_sources([dict(name="svm", header=hdr, type="user_examples::SvmSquaredHinge", id=100,
                                               wide_type="user_examples::SvmSquaredHingeWide", wide_header=hdr + "x")],
                                         str(tmp_path))
    assert len(wide) == 12 and "dispatch_wide_objective<user_examples::SvmSquaredHingeWide>" in open(wide[0]).read()
    assert hdr + "x" in open(wide[0]).read() and "UserWideRegistration" in open(wide[0]).read()
    src = open(paths[names.index("user_svm_w16.hip")]).read()
    assert "dispatch_user<16, user_examples::SvmSquaredHinge" in src and hdr in src and "UserObjectiveRegistration" in src
    assert "lbfgsb" not in src
    with pytest.raises(ValueError):
        _build.user_objective_sources([dict(name="bad", header=hdr, type="T", id=7)], str(tmp_path))
    # a functor templated over the mapping
    paths = _build.user_objective_sources([dict(name="t", header=hdr, type="ns::F<{W}, {E}>", id=101, lbfgsb=False)],
                                          str(tmp_path))
    assert "ns::F<32, 1>, ns::F<32, 2>, ns::F<32, 4>" in open(paths[2]).read()
    # the shapes of the box-constrained solver are chosen per objective: the dual SVM of the second example (n = 100)
    dual = _build.user_objective_sources([dict(name="d", header=hdr, type="ns::D<{E}>", id=101, lbfgs=False,
                                               lbfgsb=dict(n_max=128, m_max=5))], str(tmp_path))
    dn = [os.path.basename(q) for q in dual]
    assert len(dual) == 9 and "user_d_lbfgsb_exact_w16_e8_m5.hip" in dn and "user_d_lbfgsb_relaxed_w16_e8_m5.hip" in dn
    assert "ns::D<8>" in open(dual[dn.index("user_d_lbfgsb_exact_w16_e8_m5.hip")]).read()
    # the mapping table is the library's: history sizes 9, 10 and n > 128 take 32 lanes per problem
    assert _build.lbfgsb_mapping(32, 5) == (16, 2, 5) and _build.lbfgsb_mapping(100, 5) == (16, 8, 5)
    assert _build.lbfgsb_mapping(64, 8) == (16, 4, 8) and _build.lbfgsb_mapping(64, 9) == (32, 2, 10)
    assert _build.lbfgsb_mapping(100, 6) == (32, 4, 10) and _build.lbfgsb_mapping(200, 5) == (32, 8, 5)
    full = _build.lbfgsb_user_kernels(dict(n_max=256, m_max=10, hager_zhang=True))
    assert ("hz", 32, 8, 10) in full and ("relaxed", 16, 4, 8) in full and ("relaxed", 32, 1, 10) not in full


def test_dual_svm_twins_against_the_reference_binary():
    """The second worked user objective (examples/user_objective_svm_dual: the dual SVM of the reference's
    src/examples/svm_dual_lbfgsb.cc:36-77, dense 100-dimensional, box [0, C], `Lbfgsb<SvmDualObjective>` from alpha = 0):
    the exact twin equals the reference's Lbfgsb on the example's functor bit for bit; the relaxed twin (and the device
    order of the exact one) is within 1e-6 under tight stopping."""
    import ref_lib as R
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built")
    X, y = svm_data.standardised_blobs(100, 4, seed=7)
    params, Q = svm_data.dual_params(X, y)
    n, C = 100, 1.0
    lo, hi = np.zeros(n), np.full(n, C)
    x0 = np.vstack([np.zeros(n), np.random.default_rng(3).uniform(0.0, C, size=(7, n))])
    tight = O.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8, past=0)
    for st in (O.lbfgsb_default_stop(), tight):
        xr, fr, gr, pr = R.lbfgsb_minimize_batch("svm_dual", x0, m=5, stop=st, lower=lo, upper=hi, params=params)
        xe, fe, ge, pe = O.lbfgsb_minimize_batch("svm_dual", x0, m=5, stop=st, lower=lo, upper=hi, params=params,
                                                 std_sort_order=True)
        np.testing.assert_array_equal(xe, xr)
        np.testing.assert_array_equal(fe, fr)
        np.testing.assert_array_equal(ge, gr)
        np.testing.assert_array_equal(pe["num_iterations"], pr["num_iterations"])
        np.testing.assert_array_equal(pe["status"], pr["status"])
    assert np.all(pr["status"] != 1)
    # the dual solution: feasible, some multipliers at the bound, the recovered classifier separates the blobs
    assert np.all(xr >= 0) and np.all(xr <= C) and np.any(xr == C) and np.any(xr == 0)
    w = (xr[0] * y) @ X
    assert np.mean(np.sign(X @ w) == y) >= 0.9
    xf, ff, gf, pf = O.lbfgsb_fast_minimize_batch("svm_dual", x0, m=5, stop=tight, lower=lo, upper=hi, params=params)
    xb, fb, gb, pb = O.lbfgsb_minimize_batch("svm_dual", x0, m=5, stop=tight, lower=lo, upper=hi, params=params,
                                             reduction="butterfly", width=128)
    for xa, fa in ((xf, ff), (xb, fb)):
        assert np.max(np.abs(xa - xr)) <= 1e-6 and np.max(np.abs(fa - fr)) <= 1e-6


def test_the_example_library_exports_the_abi_and_is_a_separate_build():
    from cppnumericalsolvers_amd import _build, capi
    path = os.path.join(_build.PKG_DIR, "libmi355_lbfgs_svm.so")
    if not os.path.exists(path):
        pytest.skip("example library not built (run __graft_entry__.build())")
    import ctypes
    L = ctypes.CDLL(path)
    for name in capi.EXPORTED_SYMBOLS:
        assert hasattr(L, name), name


def test_user_term_unit_is_generated(tmp_path):
    """MI355_AL_TERM_USER: one generated unit holds the augmented-Lagrangian kernels for the set of term functors of the
    build, for the mappings of the dimensions asked for."""
    from cppnumericalsolvers_amd import _build
    hdr = os.path.join(ROOT, "examples", "user_al_terms", "hs_terms.hpp")
    users = [dict(name="hs024_objective", header=hdr, type="user_examples::Hs024Objective", id=100, al_term=True, objective=False),
             dict(name="ellipse", header=hdr, type="user_examples::Hs029Ellipse", id=102, al_term=True, objective=False),
             dict(name="svm", header=hdr, type="ns::Svm", id=105)]
    assert _build.al_mappings((2,)) == [(8, 1), (16, 1)]
    assert _build.al_mappings((40, 200)) == [(16, 4), (32, 2), (64, 4)]
    (path,) = _build.user_al_source(users, (2,), str(tmp_path))
    src = open(path).read()
    assert "struct UserTermsFor<8, 1>" in src and "struct UserTermsFor<16, 1>" in src and "UserTermsFor<32, 2>" not in src
    assert "TermList<UserTerm<100, user_examples::Hs024Objective>, UserTerm<102, user_examples::Hs029Ellipse>>" in src
    assert "registration_al_terms(AlLaunchTable<UserTermsFor>::table(), {100, 102}, {TermParamsFromProblem<" in src and "ns::Svm" not in src
    # term-only functors get no solver units; the plain user objective still does
    assert len(_build.user_objective_sources(users, str(tmp_path))) == 4 + 7   # (+ the Lbfgsb kernels and their table)
    with pytest.raises(ValueError):
        _build.user_al_source(users, (), str(tmp_path))
    assert _build.user_al_source(users[2:], (2,), str(tmp_path)) == []


def test_the_user_term_library_is_a_separate_build_with_the_same_abi():
    from cppnumericalsolvers_amd import _build, capi
    path = os.path.join(_build.PKG_DIR, "libmi355_lbfgs_hs.so")
    if not os.path.exists(path):
        pytest.skip("example library not built (run __graft_entry__.build())")
    import ctypes
    L = ctypes.CDLL(path)
    for name in capi.EXPORTED_SYMBOLS:
        assert hasattr(L, name), name


@pytest.mark.parametrize("d", [400, 700])
def test_svm_twin_with_hundreds_of_features_equals_the_reference(d):
    """The reference example is dynamic in the dimension; above n = 256 the device runs it on the workgroup kernel with the
    functor of examples/user_objective_svm/svm_squared_hinge_wide.hpp.  Here: the twin in the reference's order against
    the reference's Lbfgs on the example's functor, bit for bit, at n = d + 1; and the `strided` policy (the device's
    summation order in that regime) within 1e-6 of it."""
    if not ref_lib.available() or not hasattr(ref_lib.lib(), "ref_svm_minimize_batch"):
        pytest.skip("oracle/_ref/libref.so without the SVM entry")
    X, y = svm_data.two_blobs(N=120, d=d, seed=d, separation=0.15)
    p = svm_data.params(X, y, C=0.5)
    n = d + 1
    x0 = np.vstack([np.zeros(n), 0.05 * np.random.default_rng(d).normal(size=(3, n))])
    for stop in (O.default_stop(), O.parity_stop()):
        xs, fs, gs, ps = O.minimize_batch("svm_squared_hinge", x0, m=10, stop=stop, params=p)
        xr, fr, gr, pr = ref_lib.svm_minimize_batch(p, x0, m=10, stop=stop)
        np.testing.assert_array_equal(xs, xr)
        np.testing.assert_array_equal(fs, fr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(ps[k], pr[k])
    xw, fw, _, pw = O.minimize_batch("svm_squared_hinge", x0, m=10, stop=O.parity_stop(), params=p, reduction="strided", width=256)
    assert np.all(ps["status"] != 1)
    assert np.max(np.abs(xw - xs)) <= 1e-6 and np.max(np.abs(fw - fs)) <= 1e-6
