#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains <pattern>, last <n> dispatches:
    python tools/db_dispatches.py <results.db> <pattern> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f'pragma table_info({kd})')]
gx = 'd.grid_size_x' if 'grid_size_x' in cols else ('d.grid_x' if 'grid_x' in cols else '0')
wx = 'd.workgroup_size_x' if 'workgroup_size_x' in cols else '1'
rows = cur.execute(f'select s.kernel_name, d.start, d.end - d.start, {gx}, {wx} from {kd} d join {ks} s on d.kernel_id = s.id '
                   f'where s.kernel_name like ? order by d.start', (f'%{pat}%',)).fetchall()
for name, st, dur, g, w in rows[-n:]:
    print(f'{dur / 1e3:9.1f} us  grid {g:>9} wg {w:>4}  {name[:90]}')
