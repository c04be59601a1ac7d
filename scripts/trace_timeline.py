#!/usr/bin/env python3
"""Per-frame timeline of a rocprofv3 --kernel-trace CSV: splits the trace into frames at a marker kernel, then prints for
the LAST frame of the run each kernel's start offset, duration and the gap in front of it, plus per-kernel-name totals.
  python scripts/trace_timeline.py gpurun_out/r3a/t672/t_kernel_trace.csv [--frames 11] [--detail]"""
import csv, sys, collections, re
path = sys.argv[1]
detail = "--detail" in sys.argv
rows = list(csv.DictReader(open(path)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)) for r in rows))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]
# frames: a gap > 200 us between kernels separates frames (the host synchronises per frame)
frames, cur = [], [ks[0]]
for a, b in zip(ks, ks[1:]):
    if b[0] - a[1] > 150_000:
        frames.append(cur); cur = []
    cur.append(b)
frames.append(cur)
print(f"{len(ks)} kernels, {len(frames)} frames (by >150us gaps)")
for i, f in enumerate(frames[-14:]):
    span = (f[-1][1] - f[0][0]) / 1e3
    busy = sum(e - s for s, e, *_ in f) / 1e3
    print(f"frame[-{len(frames[-14:]) - i}]: {len(f):4d} kernels  span {span:8.1f} us  busy {busy:8.1f} us  gaps {span - busy:7.1f} us")
f = frames[-2]
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = f[0][0]
for s, e, n, g, w in f:
    t = tot[short(n)]
    t[0] += 1; t[1] += (e - s) / 1e3; t[2] += max(0, s - prev_end) / 1e3
    if detail:
        print(f"  +{(s - f[0][0]) / 1e3:8.1f} us  gap {max(0, s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:7.1f}  grid {g // max(w, 1):5d}x{w:<4d} {short(n)}")
    prev_end = e
print("per kernel (one frame): count, total us, gap-before us")
for n, (c, d, gp) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:60s} {c:4d} {d:8.1f} {gp:7.1f}")
