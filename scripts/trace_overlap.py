#!/usr/bin/env python3
"""How two resident batches in flight share the chip: from a rocprofv3 kernel trace of `bench.py --overlap 2` (steady-state tail), per
kernel name the launch-to-end time of a launch while ANOTHER queue's kernel is running vs alone, the fraction of the wall time with 0 / 1 / 2
kernels in flight, and the sum of kernel durations against the wall time.   python scripts/trace_overlap.py CSV [tail_kernels]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else len(ks)
ks = ks[-n_tail:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:60]


t0, t1 = ks[0][0], max(k[1] for k in ks)
ev = []
for s, e, n, q in ks:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
wall = t1 - t0
print(f"{len(ks)} kernels over {wall / 1e6:.2f} ms; sum of durations {sum(e - s for s, e, _, _ in ks) / 1e6:.2f} ms; queues: {sorted(set(k[3] for k in ks))}")
print("wall time with n kernels in flight: " + ", ".join(f"{n}: {100.0 * v / wall:.1f} %" for n, v in sorted(hist.items())))
# per kernel: duration when its whole life overlaps no other kernel vs when it does
alone, shared = collections.defaultdict(list), collections.defaultdict(list)
starts = sorted((s, e) for s, e, _, _ in ks)
import bisect
S = [s for s, _ in starts]
for s, e, n, q in ks:
    i = bisect.bisect_left(S, e)
    ov = 0
    for s2, e2 in starts[max(0, i - 40):i]:
        if (s2, e2) != (s, e) and s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
    (shared if ov > 0.2 * (e - s) else alone)[short(n)].append((e - s) / 1e3)
print(f"{'kernel':60s} {'alone: n':>9s} {'avg us':>8s} {'shared: n':>10s} {'avg us':>8s}")
for n in sorted(set(alone) | set(shared), key=lambda n_: -(sum(alone[n_]) + sum(shared[n_]))):
    a, b = alone[n], shared[n]
    if len(a) + len(b) < 20:
        continue
    print(f"{n:60s} {len(a):9d} {sum(a) / max(1, len(a)):8.1f} {len(b):10d} {sum(b) / max(1, len(b)):8.1f}")
