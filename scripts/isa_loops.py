#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in hipcc's -S output (gfx950): for every backward branch, the instructions
between its target label and the branch, by class.  Usage: python scripts/isa_loops.py file.s <kernel-name substring>"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^_Z\S*:", l) and want in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
labels, body = {}, []
for i in range(start, end):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(body)
    elif l and not l.startswith((";", ".")) and not l.endswith(":"):
        body.append(l.split(";")[0].strip())


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith(("global_load", "buffer_load")): return "vmem_load"
    if op.startswith(("global_store", "buffer_store")): return "vmem_store"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_accvgpr"): return "accvgpr_mov"
    if op.startswith("v_"): return "valu"
    return "other"


print(f"kernel lines {start}..{end}: {len(body)} instructions, {sum(1 for b in body if b.startswith('v_mfma'))} MFMAs")
for i, ins in enumerate(body):
    m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", ins)
    if not m:
        continue
    tgt = labels.get(m.group(1) or m.group(2))
    if tgt is None or tgt > i:
        continue
    seg = body[tgt:i + 1]
    mix = collections.Counter(cls(s.split()[0]) for s in seg)
    if mix["mfma"] == 0 and len(seg) < 40:
        continue
    ops = collections.Counter(s.split()[0] for s in seg if cls(s.split()[0]) == "valu")
    print(f"loop [{tgt}..{i}] {len(seg)} instr: " + ", ".join(f"{k} {v}" for k, v in mix.most_common()))
    print("    valu: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
