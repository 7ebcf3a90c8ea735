"""Summarises a ThreadSanitizer log of `make -C tests/cpp tsan-run`: for every report, where the two racing accesses
themselves are (frame #0 of each access stack), and whether any of them is in code of this repository (an
instrumented translation unit) rather than inside the uninstrumented ROCm runtime.

  python scripts/tsan_summary.py gpurun_out/<dir>/tsan.log"""
import collections
import re
import sys

text = open(sys.argv[1], errors="replace").read()
reports = re.split(r"={18}\n(?=WARNING: ThreadSanitizer)", text)
reports = [r for r in reports if r.startswith("WARNING: ThreadSanitizer")]
ours = re.compile(r"/root/repo/|cppoptlib::|mi355")
kinds = collections.Counter()
access_sites = collections.Counter()
in_repo = []
for r in reports:
    kinds[r.splitlines()[0].split("(pid")[0].strip()] += 1
    # access stacks: the blocks that start with "  Write of size", "  Atomic read of size", "  Previous write of size" ...
    for m in re.finditer(r"^  (?:Previous )?(?:[Aa]tomic )?(?:[Ww]rite|[Rr]ead) of size \d+.*?\n((?:    #\d+ .*\n)+)", r, re.M):
        frames = m.group(1).splitlines()
        # frame #0 of an access made through an intercepted libc / C++ runtime call is TSan's interceptor
        # (compiler-rt/lib/tsan): the code that made the access is the first frame after it
        site = next((f for f in frames if "compiler-rt/lib/tsan" not in f), frames[-1])
        mod = re.search(r"\(([^ ()]+?)\+0x[0-9a-f]+\)", site)
        where = mod.group(1) if mod else site.strip()[:100]
        if ours.search(site) or (mod and ("libmi355" in mod.group(1) or "_test" in mod.group(1))):
            in_repo.append(site.strip())
            where = "THIS REPOSITORY: " + where
        access_sites[where] += 1
print("%d ThreadSanitizer reports" % len(reports))
for k, v in kinds.most_common():
    print("  %4d  %s" % (v, k))
print("racing accesses (frame #0 of each access stack) by location:")
for k, v in access_sites.most_common():
    print("  %4d  %s" % (v, k))
print("accesses located in this repository's code: %d" % len(in_repo))
for s in in_repo[:20]:
    print("   ", s)
