#!/usr/bin/env python3
"""profiles/traffic.json: HBM-side bytes per launch of every sampling OP, from rocprofv3 PMC passes.

Run on the GPU box (tools/collect_traffic.sh): for each op instance of tools/bench_lift.py (self, pts,
img; bs = 2; bf16 and fp32) one pass with --pmc FETCH_SIZE and one with --pmc WRITE_SIZE (separate runs, with
--kernel-trace only, as MI355X_MICROARCH.md section HBM prescribes).  Both counters are in KB; on
gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x: fetch_corrected = 2 x raw (same guide).
Kernels are grouped into the op's forward (lift_fwd / lift_cam_fwd / lift_cam32_fwd / lift_tile_fwd / value_frags) and backward
(everything else of the lift family) and summed per launch of the op.

    python tools/make_traffic.py <dir with {op}_{FETCH_SIZE,WRITE_SIZE}_results.db> > profiles/traffic.json
"""
import collections
import json
import os
import sqlite3
import sys

OPS = {'self': 'self_attn', 'pts': 'sca_pts', 'img': 'sca_img'}
FWD = ('lift_fwd', 'lift_cam_fwd', 'lift_cam32_fwd', 'lift_tile_fwd', 'value_frags')
DTYPES = ('bf16', 'fp32')


def per_kernel(db_path, counter):
    """kernel symbol -> (mean counter total per dispatch, dispatches)"""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]      # noqa: E731
    q = f'''select s.kernel_name, sum(e.value), count(distinct d.id) from {T('rocpd_pmc_event')} e
            join {T('rocpd_info_pmc')} p on e.pmc_id = p.id
            join {T('rocpd_kernel_dispatch')} d on e.event_id = d.event_id
            join {T('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id
            where p.name = ? group by s.kernel_name'''
    return {name: (tot / nd, nd) for name, tot, nd in cur.execute(q, (counter,))}


def main():
    d = sys.argv[1]
    out = {'_note': 'HBM-side bytes per op launch from rocprofv3 PMC passes over tools/bench_lift.py (bs=2, '
                    'gfx950, per dtype): FETCH_SIZE and WRITE_SIZE in separate runs, KB -> bytes, '
                    'fetch_corrected = 2 x raw (gfx950 under-count of wide reads, MI355X_MICROARCH.md HBM '
                    'section); kernels grouped per op and pass, summed per launch of the op.',
           'commit': os.environ.get('UBV_COMMIT', 'unrecorded'), 'ops': {}, 'kernels': {}}
    for dt in DTYPES:
        ops, kernels = {}, {}
        for key, op in OPS.items():
            ff = os.path.join(d, f'{key}_{dt}_FETCH_SIZE_results.db')
            wf = os.path.join(d, f'{key}_{dt}_WRITE_SIZE_results.db')
            if not (os.path.exists(ff) and os.path.exists(wf)):
                continue
            fetch = per_kernel(ff, 'FETCH_SIZE')
            write = per_kernel(wf, 'WRITE_SIZE')
            launches = max(nd for name, (_, nd) in fetch.items() if any(k in name for k in FWD[:4]))
            agg = collections.defaultdict(lambda: [0.0, 0.0])
            for name in set(fetch) | set(write):
                if 'ubv' not in name or not any(k in name for k in ('lift_', 'value_frags', 'value_split', 'slab_reduce',
                                                                    'compact_visible', 'maps_')):
                    continue
                f, nf = fetch.get(name, (0.0, 0))
                w, nw = write.get(name, (0.0, 0))
                # per launch of the op: total over the run / op launches
                fb = 2.0 * f * 1024 * nf / launches
                wb = w * 1024 * nw / launches
                side = 'fwd' if any(k in name for k in FWD) else 'bwd'
                agg[side][0] += fb
                agg[side][1] += wb
                short = name.split('(')[0].replace('ubv::', '').replace('void ', '')[:70]
                kernels[f'{op}:{short}'] = {'fetch_corrected': int(fb), 'write': int(wb), 'per_op_launch': True}
            for side, (fb, wb) in agg.items():
                ops[f'{op}:{side}'] = {'fetch_corrected': int(fb), 'write': int(wb),
                                       'hbm_bytes_per_launch': int(fb + wb)}
        out['ops'][dt] = ops
        out['kernels'][dt] = kernels
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
