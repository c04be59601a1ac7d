#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` -> DIR/NAME_results.db).
  python scripts/rocpd_stats.py gpurun_out/vd672/vd_results.db [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, max(end-start)/1e3 from kernels "
                  "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
for r in rows[:top]:
    print(f"{r[0][:84]:84s} n={r[1]:6d} {r[2] / 1e3:9.2f} ms {100 * r[2] / tot:5.1f}%  avg {r[3]:8.1f} us  max {r[4]:8.1f} us")
