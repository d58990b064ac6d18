#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 rocpd database (default output format of ROCm 7.2).
    python tools/db_table.py <results.db> <steps incl. warmup> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f'select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {ks} s '
                   f'on d.kernel_id = s.id group by s.kernel_name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
print(f'# kernel time {tot / 1e6 / steps:.2f} ms/step, {sum(r[1] for r in rows) / steps:.0f} launches/step')
print(f'{"us/step":>9} {"pct":>6} {"calls/step":>10} {"avg_us":>9}  kernel')
for name, n, t in rows[:top]:
    print(f'{t / 1e3 / steps:9.1f} {100 * t / tot:6.2f} {n / steps:10.1f} {t / 1e3 / n:9.1f}  {name[:120]}')
