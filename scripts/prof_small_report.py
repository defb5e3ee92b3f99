#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps from a rocprofv3 kernel trace database (rocpd sqlite).
    python scripts/prof_small_report.py gpurun_out/prof_small/single_results.db"""
import collections
import re
import sqlite3
import sys

for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    rows = con.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[int(len(rows) * 0.3):]                     # the timed part of the script
    agg = collections.defaultdict(lambda: [0, 0])
    for k, s, e in rows:
        m = re.search(r"(\w+_kernel)(I[^E]*E)?", k)
        k = (m.group(1) + (m.group(2) or "")) if m else k[:50]
        agg[k][0] += 1
        agg[k][1] += e - s
    span, busy = rows[-1][2] - rows[0][1], sum(v[1] for v in agg.values())
    gaps = sorted(rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1))
    print(f"{path}: {len(rows)} dispatches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, median gap {gaps[len(gaps) // 2] / 1e3:.2f} us")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {k:44s} n={v[0]:5d} avg {v[1] / v[0] / 1e3:7.2f} us  total {v[1] / 1e6:7.2f} ms")
