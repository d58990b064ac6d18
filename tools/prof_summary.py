#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace sqlite database (rocpd) as a per-kernel table.

    python tools/prof_summary.py gpurun_out/prof/x_results.db [--top 40] [--header "text"]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--top', type=int, default=50)
    ap.add_argument('--header', default='')
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, "
        f"min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from {kd} d join {ks} s "
        f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    if a.header:
        print('# ' + a.header)
    print(f'# total kernel time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches')
    print(f'{"total_us":>12} {"pct":>6} {"calls":>6} {"avg_us":>10} {"min_us":>10} {"max_us":>10}  kernel')
    for r in rows[:a.top]:
        print(f'{r[2]:12.0f} {100 * r[2] / tot:6.2f} {r[1]:6d} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f}  {r[0][:150]}')


if __name__ == '__main__':
    main()
