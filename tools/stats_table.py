#!/usr/bin/env python3
"""Per-step table from a rocprofv3 `--kernel-trace --stats` kernel_stats.csv.
    python tools/stats_table.py <kernel_stats.csv> <steps incl. warmup> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# kernel time {tot / 1e6 / steps:.2f} ms/step, {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches/step')
print(f'{"us/step":>9} {"pct":>6} {"calls/step":>10} {"avg_us":>9}  kernel')
for r in rows[:top]:
    t = float(r['TotalDurationNs'])
    print(f'{t / 1e3 / steps:9.1f} {100 * t / tot:6.2f} {int(r["Calls"]) / steps:10.1f} '
          f'{float(r["AverageNs"]) / 1e3:9.1f}  {r["Name"][:120]}')
