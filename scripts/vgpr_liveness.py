#!/usr/bin/env python3
"""Per-instruction VGPR liveness of one kernel in a gfx950 assembly listing (hipcc -S).

usage: vgpr_liveness.py file.s [kernel-name-substring]

Builds the CFG from labels / s_branch / s_cbranch_*, runs backward liveness over physical VGPRs and
prints the pressure profile: the maximum, and the instructions around every local peak, so that the
region of the source that sets the kernel's register budget (hence waves per SIMD) can be found.
Approximations: the first vector operand of an instruction is its definition unless the mnemonic is a
store / DS write / compare / readlane; v_writelane and the permlane swaps read and write.
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    return []


NO_DEF = ("global_store", "ds_write", "ds_store", "buffer_store", "flat_store", "scratch_store", "v_cmp",
          "v_readlane", "v_readfirstlane", "s_", "global_atomic_add_x2 off", "v_cmpx")
RW = ("v_writelane", "v_permlane", "v_swap", "v_fmac", "v_mac", "v_accvgpr")


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(\w+):\s*(;.*)?$", l)
        if m and m.group(1).startswith("_Z") and want in m.group(1):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    body = []
    for l in lines[start + 1:]:
        if l.strip().startswith(".section") or l.strip().startswith(".Lfunc_end"):
            break
        body.append(l)
    insts = []      # (mnemonic, defs, uses, text)
    labels = {}
    for l in body:
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.\w+):", t)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.?\w+):$", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        vops = [(k, regs(o.split()[0] if o else "")) for k, o in enumerate(ops)]
        vops = [(k, r) for k, r in vops if r]
        defs, uses = [], []
        if vops:
            nodef = mn.startswith(NO_DEF)
            first_k, first_r = vops[0]
            if not nodef and first_k == 0:
                defs = first_r
                rest = vops[1:]
                if mn.startswith(RW):
                    uses += first_r
            else:
                rest = vops
            for _, r in rest:
                uses += r
        if "dpp" in t or "quad_perm" in t or "row_" in t:
            # DPP with bound_ctrl:0 and full masks overwrites; the old value is still an input operand
            pass
        insts.append((mn, defs, uses, t))
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (mn, _, _, t) in enumerate(insts):
        if mn == "s_endpgm":
            continue
        if mn == "s_branch":
            tgt = t.split()[1]
            succ[i].append(labels[tgt])
            continue
        if mn.startswith("s_cbranch"):
            tgt = t.split()[1]
            succ[i].append(labels[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            _, d, u, _ = insts[i]
            new = (out - set(d)) | set(u)
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    press = [len(s) for s in live_in]
    mx = max(press)
    print("instructions", n, "max live VGPRs", mx)
    # profile in coarse buckets
    step = max(1, n // 60)
    for i in range(0, n, step):
        seg = press[i:i + step]
        print("%6d  max %3d  min %3d  %s" % (i, max(seg), min(seg), insts[i][3][:70]))
    peaks = [i for i in range(n) if press[i] >= mx - 2]
    print("peak instructions (>= max-2):", len(peaks), "first", peaks[0], "last", peaks[-1])
    shown = set()
    for p in peaks[:: max(1, len(peaks) // 6)]:
        lo, hi = max(0, p - 6), min(n, p + 6)
        print("---- around", p)
        for i in range(lo, hi):
            if i not in shown:
                print("%6d %3d  %s" % (i, press[i], insts[i][3][:100]))
                shown.add(i)


if __name__ == "__main__":
    main()
