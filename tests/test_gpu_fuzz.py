"""A short draw of the randomised differential campaigns (scripts/fuzz_parity.py, scripts/fuzz_auglag.py) in every GPU run:
random solver x objective x n x m x mapping x placement x line search x arithmetic x stopping fields x boxes, and random
augmented-Lagrangian problems, device == twin compared for equality.  The long runs are in profiles/r5_fuzz_parity.txt and profiles/r6_fuzz_parity.txt."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + list(args), capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines and "summary" in lines[-1], out.stderr[-2000:]
    bad = [r for r in lines[:-1] if r.get("mismatch")]
    assert out.returncode == 0 and not bad, bad[:3]
    return lines[-1]["summary"]


def test_random_solves_equal_their_twins():
    s = _run("fuzz_parity.py", "--trials", "160", "--seed", "101")
    assert s["mismatch"] == 0 and s["compared"] >= 120 and len(s["by_solver"]) >= 6


def test_random_augmented_lagrangian_solves_equal_their_twins():
    s = _run("fuzz_auglag.py", "--trials", "60", "--seed", "102")
    assert s["mismatch"] == 0 and s["compared"] >= 50 and set(s["by_kind"]) == {"table", "family"}
