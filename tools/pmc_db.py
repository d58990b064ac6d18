#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database, per wave where it helps.
    python tools/pmc_db.py <results.db> [kernel-name substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
q = f'''select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {T('rocpd_pmc_event')} e
        join {T('rocpd_info_pmc')} p on e.pmc_id = p.id
        join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
        join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id
        where s.kernel_name like ? group by s.kernel_name, p.name'''
res = collections.defaultdict(dict)
for name, ctr, tot, nd in cur.execute(q, (f'%{sub}%',)):
    res[name][ctr] = tot / nd                      # chip-wide total per dispatch
for name, d in res.items():
    print(name[:110])
    waves = d.get('SQ_WAVES')
    for k, v in sorted(d.items()):
        per = f'   per wave {v / waves:12.1f}' if waves and k != 'SQ_WAVES' else ''
        print(f'    {k:32s} {v:16.0f}{per}')
