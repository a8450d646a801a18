#!/usr/bin/env python3
"""Instruction mix of the MFMA loop of one kernel, from the gfx950 assembly hipcc -save-temps leaves behind.
usage: tools/isa_loop_stats.py <file.s> <mangled-name-substring>  (e.g. 'beaver_gemm_group_kernelILi1ELi4ELb1ELb1ELb1ELb0ELb0ELb0E')
Prints, for the innermost basic-block run that contains the kernel's MFMAs (from the first label before the first MFMA that is
the target of a backward branch to the branch itself), the count of MFMA / VALU / SALU / vector-memory / LDS instructions."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if l.startswith("_Z") and l.split(":")[0].find(key) >= 0 and ":" in l]
    if not starts:
        sys.exit("kernel not found")
    i0 = starts[0]
    end = next(i for i in range(i0, len(txt)) if "s_endpgm" in txt[i])
    body = txt[i0:end + 1]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    print("kernel lines", len(body), "mfma", len(mf))
    # loops: backward branches
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i))
    cand = [(a, b) for a, b in loops if any(a <= x <= b for x in mf)]
    if not cand:
        print("no loop around the MFMAs"); return
    a, b = min(cand, key=lambda ab: ab[1] - ab[0])
    cls = collections.Counter()
    detail = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_mfma"): c = "mfma"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith("v_"): c = "valu"
        elif op.startswith("s_waitcnt"): c = "waitcnt"
        elif op.startswith("s_"): c = "salu"
        else: c = "other"
        cls[c] += 1
        if c == "valu":
            detail[re.sub(r"_e(32|64)$", "", op)] += 1
    print("loop lines %d..%d:" % (a, b), dict(cls))
    print("valu detail:", detail.most_common(14))


if __name__ == "__main__":
    main()
