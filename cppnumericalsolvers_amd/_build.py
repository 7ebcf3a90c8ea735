"""Build recipe of the HIP engine (in-tree, gfx950 only).

`build()` compiles cppnumericalsolvers_amd/csrc/mi355_lbfgs.hip into
cppnumericalsolvers_amd/libmi355_lbfgs.so with hipcc.  The .so is git-ignored
but travels with the tree to the GPU box.  -ffp-contract=off is part of the
numerical contract (see DESIGN.md "Arithmetic"): no FMA contraction, so the
kernels perform plain IEEE mul/add/div/sqrt and results are bit-reproducible.
"""
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmi355_lbfgs.so")
SOURCES = ["mi355_lbfgs.hip"]
HEADERS = ["lbfgs_kernel.hpp", "lbfgsb_kernel.hpp", "more_thuente_device.hpp", "objectives.hpp", "wave_primitives.hpp",
           os.path.join("..", "..", "include", "mi355_lbfgs.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built")


HASH_PATH = LIB_PATH + ".srchash"


def source_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy
    to the GPU box, contents do)."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for rel in SOURCES + HEADERS:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    return open(HASH_PATH).read().strip() != source_hash()


def build(force=False, verbose=False, extra_flags=()):
    """Compile the HIP library if it is missing or older than its sources."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + list(extra_flags) + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(HASH_PATH, "w") as f:
        f.write(source_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
