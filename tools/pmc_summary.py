#!/usr/bin/env python
"""Per-kernel mean of PMC counters from a rocprofv3 rocpd .db (rocprofv3 --pmc ... --kernel-trace)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=''):
    db = sqlite3.connect(path)
    cur = db.cursor()
    T = {re.sub(r'_[0-9a-f]{8}_.*$', '', r[0]): r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    q = f'''select s.display_name, p.name, e.value, d.id, d.end - d.start from {T['rocpd_pmc_event']} e
            join {T['rocpd_info_pmc']} p on e.pmc_id = p.id
            join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id
            join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id'''
    per = defaultdict(lambda: defaultdict(float))
    for name, pmc, val, did, dur in cur.execute(q):
        if filt in name:
            per[(re.sub(r'\s+', ' ', name)[:110], did)][pmc] += val
            per[(re.sub(r'\s+', ' ', name)[:110], did)]['~duration_ns'] = dur
    agg = defaultdict(lambda: defaultdict(list))
    for (name, did), d in per.items():
        for k, v in d.items():
            agg[name][k].append(v)
    for name, d in agg.items():
        print(name)
        for k, v in sorted(d.items()):
            print(f'    {k:32s} n={len(v):4d} mean={sum(v) / len(v):16.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
