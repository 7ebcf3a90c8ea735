#!/usr/bin/env python3
"""Builds the reference's OWN programs twice: over the reference's headers and over the drop-in headers of include/.

TEST INFRASTRUCTURE.  Runs where /root/reference exists (the authoring container); the GPU box only executes the
binaries this leaves behind.  For every entry of programs.json:

  1. the reference file is read from /root/reference and its sha256 checked (an edit names line numbers of THAT file);
  2. the edit list is applied in memory — `twin` lines, `drop-include` and (test files) `drop-test` deletions for the mi355 build only, `print` lines
     and `solver-choice` swaps for both — and the result is written to a temporary directory OUTSIDE the tree (reference text is never committed and
     never travels);
  3. `<name>_ref`   = g++ of the program over /root/reference/include + oracle/eigen_shim  -> oracle/_ref/programs/
     `<name>_mi355` = g++ of the edited program over include/ + oracle/eigen_shim, linked against
                      cppnumericalsolvers_amd/libmi355_lbfgs.so                          -> tests/cpp/_build/refprog/
     both with tests/refprog/prelude.h force-included (full-precision output; see there).

Both directories are git-ignored and travel to the GPU box as binaries, like oracle/_ref/libref.so.  `--golden` also runs
the `_ref` binaries (CPU only) and writes their outputs to tests/golden/reference_programs.json.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
REF_OUT = os.path.join(ROOT, "oracle", "_ref", "programs")
MI355_OUT = os.path.join(ROOT, "tests", "cpp", "_build", "refprog")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_programs.json")
SHIM = os.path.join(ROOT, "oracle", "eigen_shim")
PRELUDE = os.path.join(HERE, "prelude.h")
MINIGTEST = os.path.join(HERE, "minigtest")
LIBDIR = os.path.join(ROOT, "cppnumericalsolvers_amd")


def load_programs():
    with open(os.path.join(HERE, "programs.json")) as fh:
        return json.load(fh)["programs"]


def edited_source(program, build):
    """The program text for `build` ('ref' or 'mi355') as a list of lines; raises if the reference file changed."""
    path = os.path.join(REFERENCE, program["source"])
    raw = open(path, "rb").read()
    digest = hashlib.sha256(raw).hexdigest()
    if digest != program["sha256"]:
        raise RuntimeError("%s: sha256 %s, the edit list was recorded against %s" % (path, digest, program["sha256"]))
    lines = raw.decode("utf-8").split("\n")
    first, last = program.get("lines", [1, len(lines)])
    deleted, inserted = set(), {}
    for edit in program["edits"]:
        if edit["role"] == "twin" and build != "mi355":
            continue
        if edit["role"] in ("drop-include", "drop-test") and build != "mi355" and not program.get("drops_apply_to_reference_build"):
            continue
        if "delete" in edit:
            deleted.add(edit["delete"])
        if "after" in edit:     # (a `solver-choice` edit has both: the line is replaced)
            inserted.setdefault(edit["after"], []).append(edit["text"])
    out = []
    for number in range(first, last + 1):   # 1-based line numbers of the reference file
        if number not in deleted:
            out.append(lines[number - 1])
        out.extend(inserted.get(number, []))
    return out


def stale(target, sources):
    if not os.path.exists(target):
        return True
    built = os.path.getmtime(target)
    return any(os.path.getmtime(s) > built for s in sources if os.path.exists(s))


def header_files():
    found = [PRELUDE, os.path.join(HERE, "programs.json"), os.path.abspath(__file__), os.path.join(MINIGTEST, "gtest", "gtest.h")]
    for base in (os.path.join(ROOT, "include"), SHIM):
        for folder, _, names in os.walk(base):
            found.extend(os.path.join(folder, n) for n in names)
    return found


def compile_program(program, build, tmp, force=False):
    out_dir = REF_OUT if build == "ref" else MI355_OUT
    os.makedirs(out_dir, exist_ok=True)
    target = os.path.join(out_dir, "%s_%s" % (program["name"], build))
    deps = header_files() + [os.path.join(REFERENCE, program["source"])]
    library = program.get("library", "mi355_lbfgs")
    if build == "mi355":
        deps.append(os.path.join(LIBDIR, "lib%s.so" % library))
    if not force and not stale(target, deps):
        return target, False
    src = os.path.join(tmp, "%s_%s.cc" % (program["name"], build))
    with open(src, "w") as fh:
        fh.write("\n".join(edited_source(program, build)) + "\n")
    cmd = ["g++", "-std=c++17", "-O2", "-include", PRELUDE, "-I", SHIM]
    if program.get("gtest"):      # one of the reference's unit-test files: GoogleTest's interface from tests/refprog/minigtest
        cmd += ["-I", MINIGTEST] + (["-DMINIGTEST_MAIN"] if program.get("gtest_main") else [])
    # (the SVM programs include "src/examples/iris_data.h": the reference root goes AFTER the headers under test, and it has
    #  no cppoptlib/ directory of its own, so it only ever resolves that data header)
    extra = ["-I", REFERENCE] if program.get("reference_root_on_include_path") else []
    if build == "ref":
        cmd += ["-I", os.path.join(REFERENCE, "include")] + extra + [src, "-o", target]
    else:
        cmd += ["-Wall", "-Wextra", "-I", os.path.join(ROOT, "include")] + extra + [src, "-L", LIBDIR, "-l" + library, "-L/opt/rocm/lib",
                "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", target]
    done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if done.returncode != 0:
        # (compiler messages quote source lines: keep reference text out of logs that may be committed)
        raise RuntimeError("%s (%s build) does not compile:\n%s" % (program["name"], build, done.stdout[-6000:]))
    return target, True


def build_all(force=False, verbose=True):
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("/root/reference is not here: the reference programs are built in the authoring container only")
    built = {}
    with tempfile.TemporaryDirectory(prefix="refprog_") as tmp:
        for program in load_programs():
            for build in ("ref", "mi355"):
                target, fresh = compile_program(program, build, tmp, force)
                built[(program["name"], build)] = target
                if verbose and fresh:
                    print("built", os.path.relpath(target, ROOT))
    return built


def write_golden(only=None):
    """Outputs of the `_ref` binaries (the reference's headers, CPU): the numbers the mi355 builds are held against.
    `only`: re-run just these programs and keep the recorded outputs of the others (svm_primal_al runs its 10,001 outer
    iterations for 8.5 minutes over the eager Eigen stand-in)."""
    golden = {"_generator": "tests/refprog/build_refprogs.py --golden", "programs": {}}
    if only and os.path.exists(GOLDEN):
        with open(GOLDEN) as fh:
            golden["programs"] = json.load(fh)["programs"]
    for program in load_programs():
        if only and program["name"] not in only and program["name"] in golden["programs"]:
            continue
        binary = os.path.join(REF_OUT, program["name"] + "_ref")
        done = subprocess.run([binary], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              timeout=1800 if program.get("slow_reference") else 120)
        if done.returncode != 0:
            raise RuntimeError("%s exited with %d" % (binary, done.returncode))
        golden["programs"][program["name"]] = {"scalar": program["scalar"], "stdout": done.stdout.split("\n")}
    order = [p["name"] for p in load_programs()]
    golden["programs"] = {name: golden["programs"][name] for name in order if name in golden["programs"]}
    with open(GOLDEN, "w") as fh:
        json.dump(golden, fh, indent=1)
        fh.write("\n")
    print("wrote", os.path.relpath(GOLDEN, ROOT))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--golden", action="store_true", help="also run the _ref binaries and rewrite tests/golden/reference_programs.json")
    ap.add_argument("--only", nargs="*", help="with --golden: re-run only these programs, keep the others' recorded outputs")
    args = ap.parse_args()
    build_all(force=args.force)
    if args.golden:
        write_golden(args.only)
    sys.exit(0)
