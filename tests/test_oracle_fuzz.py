"""A short draw of scripts/fuzz_oracle_vs_reference.py in every CPU run: random solver x objective x n x m x line search x
stopping fields x boxes, ridge data, condition_hessian thresholds, random augmented-Lagrangian problems — the sequential twin
== the reference binary (oracle/_ref/libref.so), compared for equality.  The long runs: profiles/r5_fuzz_oracle_vs_reference.txt."""
import json
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so not available")
def test_random_twin_solves_equal_the_reference_binary():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_oracle_vs_reference.py"), "--trials", "250", "--seed", "77"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines and "summary" in lines[-1], out.stderr[-2000:]
    bad = [r for r in lines[:-1] if r.get("mismatch")]
    s = lines[-1]["summary"]
    assert out.returncode == 0 and not bad and s["mismatch"] == 0, bad[:3]
    assert s["compared"] == 250 and len(s["by_kind"]) == 7
