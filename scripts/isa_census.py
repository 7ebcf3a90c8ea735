"""Instruction-class census of one kernel from a gfx950 assembly listing (hipcc -S --cuda-device-only).

  python scripts/isa_census.py <listing.s> <substring of the mangled kernel name> [--phases]

Buckets every instruction of the kernel's ITERATION LOOP (the outermost loop: `Loop: Header=... Depth=1` and everything
nested in it) and of the whole kernel.  With --phases the listing must come from a -DMI355_LBFGS_PHASE_TIMING (or
-DMI355_LBFGSB_PHASE_TIMING) build: the phase markers read the cycle counter (s_memtime), so the loop body is cut at
every s_memtime and the census is printed per segment, in program order (the legend is the order of the MI355_LPHASE /
MI355_PHASE markers in the kernel source)."""
import collections
import re
import sys

CLASSES = ["fp64 fma/mul/add", "fp64 div/rcp/sqrt sequence", "fp64 max/min/cmp", "dpp move", "v_cndmask", "v_mov",
           "v_readlane/writelane (SGPR spill traffic)", "permlane/bpermute-side valu", "integer / address valu",
           "other valu", "lds", "global / scratch", "salu", "s_waitcnt / s_nop", "branch"]


def classify(line):
    op = line.split()[0]
    if "dpp" in line or "quad_perm" in line or " row_" in line:
        return "dpp move"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")):
        return "fp64 fma/mul/add"
    if op.startswith(("v_div_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_ldexp", "v_frexp", "v_trig")):
        return "fp64 div/rcp/sqrt sequence"
    if op.startswith(("v_max_f64", "v_min_f64", "v_cmp", "v_cmpx")):
        return "fp64 max/min/cmp"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "v_readlane/writelane (SGPR spill traffic)"
    if op.startswith("v_permlane"):
        return "permlane/bpermute-side valu"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith(("v_add_u", "v_add_co", "v_sub", "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_mad_",
                      "v_mul_lo", "v_mul_hi", "v_mul_u", "v_bfe", "v_ffb", "v_min_", "v_max_", "v_add3", "v_not",
                      "v_bcnt", "v_mbcnt", "v_alignbit", "v_perm")):
        return "integer / address valu"
    if op.startswith("v_"):
        return "other valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "global / scratch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "s_waitcnt / s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return None


def is_valu(c):
    return c in CLASSES[:10]


def kernel_lines(path, key):
    lines = open(path).read().splitlines()
    start = end = None
    for i, l in enumerate(lines):
        if start is None and key in l and not l.startswith(("\t", ".", ";")) and re.match(r"^[A-Za-z_][\w$.]*:", l):
            start = i
        if start is not None and l.strip().startswith("s_endpgm"):
            end = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + key)
    return lines[start:end + 1]


def census(body):
    c = collections.Counter()
    for l in body:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        k = classify(s)
        if k:
            c[k] += 1
    return c


def show(title, c):
    valu = sum(v for k, v in c.items() if is_valu(k))
    print("%s   (VALU %d, all %d)" % (title, valu, sum(c.values())))
    for k in CLASSES:
        if c[k]:
            share = (" %5.1f %% of VALU" % (100.0 * c[k] / valu)) if is_valu(k) and valu else ""
            print("   %-44s %6d%s" % (k, c[k], share))


def main():
    path, key = sys.argv[1], sys.argv[2]
    phases = "--phases" in sys.argv
    lines = kernel_lines(path, key)
    show("whole kernel", census(lines))
    # the iteration loop: from the first depth-1 loop header to the last line that says it belongs to a loop
    first = next(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l)
    last = max(i for i, l in enumerate(lines) if "in Loop: Header=" in l or "Parent Loop" in l or "Inner Loop Header" in l)
    while last + 1 < len(lines) and not lines[last + 1].strip().endswith(":"):
        last += 1
    loop = lines[first:last + 1]
    print()
    show("iteration loop (static: every instruction once, whatever its trip count or divergence)", census(loop))
    if phases:
        print()
        segs, cur = [], []
        for l in loop:
            cur.append(l)
            if l.strip().startswith("s_memtime"):
                segs.append(cur)
                cur = []
        segs.append(cur)
        for i, s in enumerate(segs):
            show("segment %d (up to the %s s_memtime)" % (i, ["1st", "2nd", "3rd"][i] if i < 3 else "%dth" % (i + 1)), census(s))
    meta = [l.strip() for l in open(path).read().splitlines() if re.search(r"\.(vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size):", l)]
    print()
    print("metadata (all kernels of the listing): " + " ".join(meta))


if __name__ == "__main__":
    main()
