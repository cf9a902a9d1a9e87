#!/usr/bin/env python3
"""Instruction classes per basic block of one kernel in hipcc's gfx950 assembly (-save-temps .s file).

    hipcc --offload-arch=gfx950 -O3 ... --save-temps=obj -c kernels.hip -o /tmp/dis/kernels.o
    python tools/isa_table.py /tmp/dis/kernels-hip-amdgcn-amd-amdhsa-gfx950.s \
        'k_jac_fusedINS_5ModelILj511ELb0ELb0EEELb1ELb0ELi3' [block=count ...]

Without block=count arguments: one line per basic block (label, instruction counts by class).  With them (e.g.
LBB5_109=1 LBB5_120=7 LBB5_119=7): the weighted total, i.e. the instructions one trip through those blocks issues.
Classes: fp64 (FP64 vector arithmetic), mfma, valu (every other vector ALU instruction, DPP moves included), ds_read /
ds_write (with the bytes per lane they move), vmem, smem, salu, waitcnt, branch, other.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if re.search(r"_f64", op) and not op.startswith("v_cvt") and not op.startswith("v_cmp"):
            return "fp64"
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ds_write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setprio", "s_nop", "s_sleep")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def ds_bytes(op):
    m = re.search(r"_b(\d+)$", op)
    return int(m.group(1)) // 8 if m else 0


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    weights = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        weights[k] = float(v)
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", ln))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip() == "s_endpgm" and
               i + 1 < len(lines) and lines[i + 1].startswith(".Lfunc_end"))
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = collections.Counter()
    for ln in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1).lstrip(".")
            blocks.setdefault(cur, collections.Counter())
            continue
        m = re.match(r"^; %bb\.(\d+):", ln)
        if m:
            cur = "bb." + m.group(1)
            blocks.setdefault(cur, collections.Counter())
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        blocks[cur][c] += 1
        if c in ("ds_read", "ds_write"):
            blocks[cur][c + "_bytes"] += ds_bytes(op)
    cols = ["fp64", "mfma", "valu", "ds_read", "ds_read_bytes", "ds_write", "ds_write_bytes", "vmem", "smem", "salu", "waitcnt",
            "branch", "other"]
    print("%-12s " % "block" + " ".join("%8s" % c[:8] for c in cols))
    total = collections.Counter()
    for name, cnt in blocks.items():
        if weights and name not in weights:
            continue
        w = weights.get(name, 1.0)
        print("%-12s " % (name + ("" if not weights else " x%g" % w)) + " ".join("%8g" % (cnt[c]) for c in cols))
        for c in cols:
            total[c] += w * cnt[c]
    if weights:
        print("%-12s " % "weighted" + " ".join("%8g" % total[c] for c in cols))


if __name__ == "__main__":
    main()
