#!/usr/bin/env python3
"""Per-kernel totals of the trailing steady-state stretch of a rocprofv3 kernel trace, normalised per incremental frame
(36 select launches = one ViTDet frame).  python scripts/trace_summary.py CSV [tail_kernels]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)) for r in rows))
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else len(ks)
ks = ks[-n_tail:]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
sel = sum(1 for k in ks if "select_kernel" in k[2])
frames = sel / 36.0
if frames < 1:   # no selection launches in the trace: count the global blocks' preparation launches (4 per frame)
    frames = sum(1 for k in ks if "stream_prep_kernel" in k[2] or "rel_terms" in k[2]) / 4.0
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
durs = collections.defaultdict(list)
prev = ks[0][0]
for s, e, n, g, w in ks:
    t = tot[short(n)]
    t[0] += 1; t[1] += (e - s) / 1e3; t[2] += min(max(0, s - prev), 50_000) / 1e3
    durs[short(n)].append((e - s) / 1e3)
    prev = max(prev, e)
busy = sum(v[1] for v in tot.values()); gaps = sum(v[2] for v in tot.values())
print(f"{len(ks)} kernels, ~{frames:.1f} gated frames; per frame: busy {busy / frames:.1f} us, gaps {gaps / frames:.1f} us")
print(f"{'kernel':70s} {'n/frame':>8s} {'us/frame':>9s} {'avg us':>8s} {'gap/frame':>9s}  p10 / p50 / p90 us")
for n, (c, d, gp) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    v = sorted(durs[n])
    pct = lambda q: v[min(len(v) - 1, int(q * len(v)))]   # a kernel launched in several roles (shapes) shows up as a spread
    print(f"{n:70s} {c / frames:8.1f} {d / frames:9.1f} {d / c:8.1f} {gp / frames:9.1f}  {pct(0.1):5.1f} / {pct(0.5):5.1f} / {pct(0.9):5.1f}")
