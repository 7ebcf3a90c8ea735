#!/usr/bin/env python
"""Register / scratch budget of every kernel of the built library, from the code-object metadata of the objects under
cppnumericalsolvers_amd/_build/ (no GPU needed):  python scripts/kernel_resources.py [> profiles/<tag>_kernel_resources.txt]

For each object: dump the .hip_fatbin section, unbundle the gfx950 code object, read the amdhsa.kernels notes."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "k.co")
    if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "x.o")],
                      capture_output=True).returncode != 0 or not os.path.exists(fat):
        return []
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + fat, "--list"], capture_output=True, text=True)
    target = next((t for t in r.stdout.split() if "gfx950" in t), None)
    if not target:
        return []
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + fat, "--targets=" + target, "--output=" + co,
                    "--unbundle"], check=True, capture_output=True)
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\.%s:\s+(\S+)" % name, block)
            return m.group(1) if m else "?"
        agpr = block.split()[0]
        out.append(dict(name=field("name"), vgpr=field("vgpr_count"), agpr=agpr, sgpr=field("sgpr_count"),
                        sgpr_spill=field("sgpr_spill_count"), vgpr_spill=field("vgpr_spill_count"),
                        scratch=field("private_segment_fixed_size"), lds=field("group_segment_fixed_size")))
    os.remove(fat)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [l.replace("mi355::", "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            for l in r.stdout.splitlines()]


def main():
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cppnumericalsolvers_amd", "_build")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(build, "*.o"))):
            for k in kernels_of(obj, tmp):
                k["unit"] = os.path.basename(obj)[:-2]
                rows.append(k)
    for k, d in zip(rows, demangle([k["name"] for k in rows])):
        k["short"] = d
    rows.sort(key=lambda k: (-int(k["scratch"]), k["unit"], k["short"]))
    print("kernels: %d; with scratch: %d; units: %d  (vgpr counts include AGPRs used as spill space only when agpr > 0)"
          % (len(rows), sum(int(k["scratch"]) > 0 for k in rows), len({k["unit"] for k in rows})))
    print("%-28s %5s %5s %5s %6s %6s %8s  %s" % ("unit", "vgpr", "agpr", "sgpr", "s-spl", "v-spl", "scratchB", "kernel"))
    for k in rows:
        print("%-28s %5s %5s %5s %6s %6s %8s  %s" % (k["unit"][:28], k["vgpr"], k["agpr"], k["sgpr"], k["sgpr_spill"],
                                                      k["vgpr_spill"], k["scratch"], k["short"][:150]))


if __name__ == "__main__":
    main()
