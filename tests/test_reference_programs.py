"""The reference's OWN programs over the drop-in headers (round-5 verdict, "Next" 1).

src/examples/simple.cc, linear_regression.cc, constrained_simple2.cc and the three README programs (quick start, ridge
expression templates, constrained) are built twice by tests/refprog/build_refprogs.py — over the reference's headers
(`<name>_ref`, CPU, the checker) and over include/ with the recorded edit list of tests/refprog/programs.json
(`<name>_mi355`, the whole solve on the GPU).  The edit list is the proof of "one device-twin line per functor class":
the CPU tests below hold it to exactly that; the GPU tests run the programs and compare every number they print with
the reference build's at 1e-6 (5e-4 for the float program, whose reference build is itself only that close to the
analytic optimum).

Since the end of round 6 the same machinery builds the reference's own UNIT-TEST files — src/test/cstep_test.cc (7 tests,
no edit at all), src/test/verify.cc (its Bfgs / Lbfgs / Lbfgsb / finite-difference / constrained tests: 9; the tests of the
four solvers outside SURVEY section 8 dropped) and src/test/augmented_lagrangian_test.cc (all 27) — over include/, with
GoogleTest's interface from tests/refprog/minigtest (GoogleTest is not in the image), and runs them on the GPU: 43 of the
reference's own tests pass on the device through the drop-in headers.  src/test/hager_zhang_test.cc is not among them: its
1-D test functions (a cubic, a quartic, a quadratic with a linear term as OBJECTIVES of a stand-alone search) have no
kernel in the closed objective menu; tests/cpp/hager_zhang_test.cc restates its two quadratic cases over the menu.

Reference text is never stored: edits name line numbers of files identified by sha256, the edited sources live in a
temporary directory outside the tree, only binaries and the reference build's OUTPUT (tests/golden/reference_programs.json)
are kept.
"""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "refprog"))
import build_refprogs as rp  # noqa: E402

PROGRAMS = rp.load_programs()
NAMES = [p["name"] for p in PROGRAMS]
# one record builder, optionally offset by a constant as the functor's own return statement is (`x(0) - 0.5`, `2 - (...)`)
TWIN_LINE = re.compile(r"^  auto DeviceTwin\(\) const \{ return ([0-9.]+ - )?cppoptlib::mi355::twin::[A-Za-z]+\(.*\)"
                       r"( - (lower_bound|[0-9.]+))?; \}$")
# the reference's TEST files hold functors that are sums of two menu primitives or carry their constant in a member
# (`x0 - target`, `bound - x0`): still ONE line and ONE return statement, an expression over record builders
TWIN_LINE_OF_A_TEST_FILE = re.compile(r"^  auto DeviceTwin\(\) const \{ return [^;]*cppoptlib::mi355::twin::[A-Za-z]+\([^;]*\)[^;]*; \}$")
HAVE_REFERENCE = os.path.isdir(rp.REFERENCE)
# solvers outside SURVEY section 8 (the only includes an edit list may drop)
NON_SECTION8_SOLVERS = ("conjugated_gradient_descent.h", "gradient_descent.h", "nelder_mead.h", "newton_descent.h",
                        "trust_region_newton.h")


def test_edit_lists_hold_one_twin_line_per_functor_class_and_nothing_else():
    for program in PROGRAMS:
        classes = []
        for edit in program["edits"]:
            assert edit["role"] in ("twin", "drop-include", "print", "solver-choice") or (program.get("gtest") and edit["role"] == "drop-test"), edit
            if edit["role"] == "twin":
                assert (TWIN_LINE_OF_A_TEST_FILE if program.get("gtest") else TWIN_LINE).match(edit["text"]), edit["text"]   # one line
                assert "\n" not in edit["text"]
                classes.append(edit["class"])
            elif edit["role"] in ("drop-include", "drop-test"):
                assert set(edit) == {"role", "delete"}
            elif edit["role"] == "solver-choice":   # the file's own alternative, replacing the active line
                assert edit["delete"] == edit["after"]
                assert re.match(r"^  using Solver = cppoptlib::solver::(Bfgs|Lbfgsb)<FunctionExprXd2>;$", edit["text"])
            else:
                assert edit["text"].lstrip().startswith("std::cout <<") and "\n" not in edit["text"]
        assert len(classes) == len(set(classes)) >= (0 if program.get("gtest") else 1), program["name"]   # ONE line per class
    assert {p["name"] for p in PROGRAMS} >= {"simple", "linear_regression", "constrained_simple", "constrained_simple2",
                                             "readme_ridge"}


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference is only in the authoring container")
def test_edit_lists_match_the_reference_files():
    """Every functor class of a program got its twin line (none was left to another mechanism), the line sits inside that
    class, and the only lines deleted are #includes of solvers outside SURVEY section 8."""
    for program in PROGRAMS:
        lines = open(os.path.join(rp.REFERENCE, program["source"]), encoding="utf-8").read().split("\n")
        first, last = program.get("lines", [1, len(lines)])
        declared = {}
        for number in range(first, last + 1):
            # (the base list may start on the next line; a test file also declares functors inside a TEST body, indented)
            m = re.match(r"^(?:  )?class (\w+)( : public |$)" if program.get("gtest") else r"^class (\w+)( : public |$)", lines[number - 1])
            if m and (m.group(2) or lines[number].lstrip().startswith(": public ")):
                if "testing::Test" in lines[number - 1]:     # a GoogleTest fixture of a test file, not a functor
                    continue
                declared[m.group(1)] = number
        twins = {e["class"]: e["after"] for e in program["edits"] if e["role"] == "twin"}
        # every functor class has its twin line — or, in a test file, is named with the reason why no solver ever sees it
        host_only = program.get("host_only_classes", {})
        assert not (set(twins) & set(host_only)) and all(host_only.values())
        assert set(twins) | set(host_only) == set(declared), (program["name"], sorted(declared), sorted(twins))
        for name, after in twins.items():
            # the line lands before the class's closing brace: the next line of the file is `};`
            assert declared[name] < after and lines[after].strip() == "};", (program["name"], name)
        for e in program["edits"]:
            if e["role"] == "drop-include":
                text = lines[e["delete"] - 1]
                assert text.startswith('#include "cppoptlib/solver/') and text.rstrip('"').endswith(NON_SECTION8_SOLVERS), text
            if e["role"] == "drop-test":    # a line that instantiates the file's typed tests on a solver outside section 8
                text = lines[e["delete"] - 1]
                assert re.match(r"^SOLVER_SETUP(_CONSERVATIVE)?\((GradientDescent|ConjugatedGradientDescent|NewtonDescent|NelderMead), ", text), text
            if e["role"] == "solver-choice":
                # the replaced line is the file's active choice, and the alternative is spelled in the file's own comments
                assert lines[e["delete"] - 1].startswith("  using Solver = cppoptlib::solver::Lbfgs<")
                solver = re.search(r"solver::(\w+)<", e["text"]).group(1)
                commented = " ".join(l.strip().lstrip("/ ") for l in lines[e["delete"] - 8:e["delete"] + 3] if l.strip().startswith("//"))
                assert ("cppoptlib::solver::%s<FunctionExprXd2>;" % solver) in commented, (solver, commented)
        # edited sources differ from the reference file by exactly the recorded lines
        assert len(rp.edited_source(program, "mi355")) == (last - first + 1) + sum(
            (1 if "after" in e else 0) - (1 if "delete" in e else 0) for e in program["edits"])
        drops = sum(1 for e in program["edits"] if e["role"] in ("drop-include", "drop-test")) if program.get("drops_apply_to_reference_build") else 0
        assert len(rp.edited_source(program, "ref")) == (last - first + 1) + sum(1 for e in program["edits"] if e["role"] == "print") - drops


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference is only in the authoring container")
def test_reference_programs_compile_over_the_drop_in_headers_with_plain_gxx():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(rp.LIBDIR, "libmi355_lbfgs.so")):
        ge.build()
    built = rp.build_all(verbose=False)
    for name in NAMES:
        for build in ("ref", "mi355"):
            assert os.access(built[(name, build)], os.X_OK)


def _numbers_and_skeleton(line):
    """A printed line as (words with numbers replaced by '#', the numbers)."""
    words, numbers = [], []
    for token in line.replace(",", " ").split():
        try:
            numbers.append(float(token))
            words.append("#")
        except ValueError:
            words.append(token)
    return words, numbers


def tolerance_of(program):
    """1e-6 (the north star's tolerance) unless the program is computed in float (5e-4) or records a looser one where the
    reference's own tests do (programs.json `tolerance`)."""
    return float(program.get("tolerance", 5e-4 if program["scalar"] == "float" else 1e-6))


def compare_outputs(name, scalar, got, want, tol=None, skip=()):
    """Line by line: the same words, the same integers, every real number within the tolerance; on lines that start with
    one of `skip` (programs.json `trajectory_dependent_lines`) only the words."""
    if tol is None:
        tol = 5e-4 if scalar == "float" else 1e-6
    got = [l for l in got if l.strip()]
    want = [l for l in want if l.strip()]
    assert len(got) == len(want), "%s: %d lines printed, the reference build prints %d\n%s" % (name, len(got), len(want), "\n".join(got))
    worst = 0.0
    for g, w in zip(got, want):
        gw, gn = _numbers_and_skeleton(g)
        ww, wn = _numbers_and_skeleton(w)
        assert gw == ww, "%s: %r vs the reference build's %r" % (name, g, w)
        if any(g.lstrip().startswith(label) for label in skip):
            continue
        # x / gradient lines of the progress printer go through a string stream of their own: six significant digits
        coarse = g.lstrip().startswith(("X:", "Gradient:"))
        for a, b in zip(gn, wn):
            if float(b).is_integer() and float(a).is_integer() and abs(b) < 1e6 and ("teration" in g):
                assert a == b, "%s: %r vs %r" % (name, g, w)
                continue
            allowed = tol + (1e-5 * abs(b) if coarse else 0.0)
            assert abs(a - b) <= allowed, "%s: %r vs the reference build's %r (|diff| %.3g > %.3g)" % (name, g, w, abs(a - b), allowed)
            worst = max(worst, abs(a - b))
    return worst


def _golden():
    with open(rp.GOLDEN) as fh:
        return json.load(fh)["programs"]


def test_golden_outputs_cover_every_program():
    golden = _golden()
    assert set(golden) == set(NAMES)
    for name in NAMES:
        if next(p for p in PROGRAMS if p["name"] == name).get("gtest"):
            # one of the reference's test files: every test of the reference-headers build passed, and there are some
            out = golden[name]["stdout"]
            ran = [l for l in out if l.startswith("[ RUN      ]")]
            assert ran and len([l for l in out if l.startswith("[       OK ]")]) == len(ran)
            assert ("[  PASSED  ] %d tests." % len(ran)) in out and not any("FAILED" in l for l in out), name
            continue
        # (every program prints its solution: argmin / x* / "Optimal x", the SVM programs their weight vector)
        assert any("argmin" in l or "x*" in l or "Optimal x" in l or l.lstrip().startswith("w:") for l in golden[name]["stdout"]), name


@pytest.mark.parametrize("name", NAMES)
def test_reference_build_reproduces_the_golden_output(name):
    """oracle/_ref/programs/<name>_ref (the reference's headers, CPU) prints what tests/golden holds: the fixture is the
    reference's own output, not something typed in."""
    binary = os.path.join(rp.REF_OUT, name + "_ref")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/programs was not built here")
    if next(p for p in PROGRAMS if p["name"] == name).get("slow_reference"):
        pytest.skip("the reference build of this program runs for minutes (programs.json `slow_reference`); its recorded output "
                    "comes from tests/refprog/build_refprogs.py --golden --only %s" % name)
    done = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    scalar = next(p["scalar"] for p in PROGRAMS if p["name"] == name)
    assert compare_outputs(name, scalar, done.stdout.split("\n"), _golden()[name]["stdout"]) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_reference_program_on_the_device_matches_the_reference_build(name):
    binary = os.path.join(rp.MI355_OUT, name + "_mi355")
    assert os.path.exists(binary), "%s did not travel to this box (tests/refprog/build_refprogs.py builds it where the reference is)" % binary
    done = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stdout + done.stderr
    program = next(p for p in PROGRAMS if p["name"] == name)
    scalar, tol, skip = program["scalar"], tolerance_of(program), tuple(program.get("trajectory_dependent_lines", ()))
    worst = compare_outputs(name, scalar, done.stdout.split("\n"), _golden()[name]["stdout"], tol, skip)
    ref = os.path.join(rp.REF_OUT, name + "_ref")
    live = ""
    if os.path.exists(ref) and not program.get("slow_reference"):   # the reference build itself, run on this box's host cores
        r = subprocess.run([ref], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0
        worst = max(worst, compare_outputs(name, scalar, done.stdout.split("\n"), r.stdout.split("\n"), tol, skip))
        live = " and its live run"
    if program.get("gtest"):
        ran = [l for l in done.stdout.split("\n") if l.startswith("[ RUN      ]")]
        passed = [l for l in done.stdout.split("\n") if l.startswith("[       OK ]")]
        assert ran and len(passed) == len(ran)
        print("\n%s: %d of the %d tests of the reference's %s pass on the device through include/ (the same tests, in the same order, "
              "as its build over the reference's headers%s)" % (name, len(passed), len(ran), program["source"], live))
        return
    print("\n%s: every printed number within %.3g of the reference build's (golden%s); tolerance %g%s%s" %
          (name, worst, live, tol, " (float program)" if scalar == "float" else "",
           "; numbers NOT compared on the trajectory-dependent lines %s" % list(skip) if skip else ""))


@pytest.mark.gpu
def test_drop_in_headers_compile_on_this_box():
    """The header-only boundary compiled once where it runs (the C++ test binaries travel prebuilt): -fsyntax-only of one
    test over include/, and of the same over an <Eigen/Core> (the CPPOPTLIB_MI355_HAVE_EIGEN branch)."""
    src = os.path.join(ROOT, "tests", "cpp", "function_expr_test.cc")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp")]
    for extra in ([], ["-I", rp.SHIM]):
        done = subprocess.run(base + extra + [src], capture_output=True, text=True, timeout=300)
        assert done.returncode == 0 and "warning" not in done.stderr, done.stderr
