#!/usr/bin/env python3
"""Instruction mix per K step of one kernel, from the gfx950 assembly `hipcc -save-temps` leaves behind.
usage: tools/isa_loop_stats.py <file.s> <mangled-name-substring> [mfma-per-step]
Counts the instructions from the kernel's first MFMA to its last loop back edge, leaving out the basic blocks that store to global
memory (the per-tile epilogues), and divides by the number of K steps in that range (MFMAs / mfma-per-step, default 36)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 36
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if l.startswith("_Z") and ":" in l and key in l.split(":")[0]]
    if not starts:
        sys.exit("kernel not found")
    i0 = starts[0]
    end = next(i for i in range(i0, len(txt)) if "s_endpgm" in txt[i])
    body = txt[i0:end + 1]
    vg = [l for l in txt if key in l and ".num_vgpr" in l]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    back = [i for i, l in enumerate(body) for m in [re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if m and m.group(1) in labels and labels[m.group(1)] < i]
    lo = min([labels[re.search(r"(\.LBB\d+_\d+)", body[i]).group(1)] for i in back if i > mf[-1]] or [mf[0]])
    hi = max([i for i in back if i > mf[-1]] or [mf[-1]])
    blocks, cur = [], []
    for i in range(lo, hi + 1):
        if re.match(r"^\.LBB\d+_\d+:", body[i]) and cur:
            blocks.append(cur); cur = []
        cur.append(body[i])
    blocks.append(cur)
    cls, detail = collections.Counter(), collections.Counter()
    for blk in blocks:
        ops = [l.strip().split()[0] for l in blk if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
        if any(o.startswith(("global_store", "global_atomic")) for o in ops):
            cls["epilogue_blocks"] += 1
            continue
        for o in ops:
            c = classify(o)
            cls[c] += 1
            if c == "valu":
                detail[re.sub(r"_e(32|64)$", "", o)] += 1
    steps = max(1, cls["mfma"] // per)
    print("vgprs:", vg[0].split(",")[-1].strip() if vg else "?", " loop lines %d..%d, %d K steps in the loop body" % (lo, hi, steps))
    print("per K step:", {k: round(v / steps, 1) for k, v in cls.items() if k != "epilogue_blocks"}, " epilogue blocks left out:", cls["epilogue_blocks"])
    print("valu detail per K step:", [(k, round(v / steps, 1)) for k, v in detail.most_common(14)])


if __name__ == "__main__":
    main()
