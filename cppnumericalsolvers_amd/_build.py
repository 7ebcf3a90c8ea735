"""Build recipe of the HIP engine (in-tree, gfx950 only).

`build()` compiles the translation units under cppnumericalsolvers_amd/csrc/ (the C-ABI plus one file
per lanes-per-problem value and one for L-BFGS-B, in parallel) and links them into
cppnumericalsolvers_amd/libmi355_lbfgs.so with hipcc.  The .so is git-ignored
but travels with the tree to the GPU box.  -ffp-contract=off is part of the
numerical contract (see DESIGN.md "Arithmetic"): no FMA contraction, so the
kernels perform plain IEEE mul/add/div/sqrt and results are bit-reproducible.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmi355_lbfgs.so")
SOURCES = ["mi355_lbfgs.hip", "dispatch_w8.hip", "dispatch_w16.hip", "dispatch_w32.hip", "dispatch_w64.hip",
           "dispatch_lbfgsb.hip", "dispatch_ridge_mfma.hip", "auglag.hip", "auglag_fused.hip", "host_pipeline.hip"]
HEADERS = ["engine_internal.hpp", "lbfgs_kernel.hpp", "lbfgsb_kernel.hpp", "more_thuente_device.hpp",
           "hager_zhang_device.hpp", "ridge_mfma_kernel.hpp", "auglag_device.hpp", "auglag_internal.hpp", "objectives.hpp", "wave_primitives.hpp",
           os.path.join("..", "..", "include", "mi355_lbfgs.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
OBJ_DIR = os.path.join(PKG_DIR, "_build")


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


def build(force=False, verbose=False, extra_flags=(), output=None):
    """Compile the HIP library if it is missing or older than its sources.  `output` builds a variant
    (other -D flags) next to the default library without touching it (scripts/ab_variants.sh)."""
    target = output or LIB_PATH
    if output is None and not force and not is_stale():
        return LIB_PATH
    hipcc = hipcc_path()
    # objects of variant builds go under _build/ too (one ignore rule keeps them off the GPU box)
    obj_dir = OBJ_DIR if output is None else os.path.join(OBJ_DIR, "variant_" + os.path.basename(output))
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-lpthread", "-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(target + ".tmp", target)
    if output is None:
        with open(HASH_PATH, "w") as f:
            f.write(source_hash() + "\n")
    return target


if __name__ == "__main__":
    print(build(force=True, verbose=True))
