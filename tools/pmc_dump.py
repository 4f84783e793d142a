#!/usr/bin/env python
"""Every counter of a rocprofv3 --pmc pass, per kernel (mean per launch), next to the mean duration:   pmc_dump.py pass.db [name-filter-regex]
SQ_* cycle counters are summed over the chip's waves / SIMDs by the profiler; the column 'per SQ_WAVE_CYCLES' (when that counter is in the
pass) relates a wave-state counter to the waves' total cycles, 'per GRBM cycle' = value / (GRBM_GUI_ACTIVE / 8 XCDs)."""
import re, sqlite3, sys
from collections import defaultdict


def main(path, flt=None):
    db = sqlite3.connect(path); cur = db.cursor()
    T = {re.sub(r'_[0-9a-f]{8}_.*$', '', r[0]): r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    q = f'''select s.display_name, p.name, e.value, d.id, d.end - d.start from {T['rocpd_pmc_event']} e
            join {T['rocpd_info_pmc']} p on e.pmc_id = p.id join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id
            join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id'''
    per = defaultdict(lambda: defaultdict(float))
    for name, pmc, val, did, dur in cur.execute(q):
        name = re.sub(r'\(.*$', '', re.sub(r'\s+', ' ', name.replace('(anonymous namespace)::', '').replace('void ', '')))
        if flt and not re.search(flt, name):
            continue
        per[(name, did)][pmc] += val
        per[(name, did)]['_dur'] = dur
    agg = defaultdict(lambda: defaultdict(float))
    for (name, did), d in per.items():
        agg[name]['_n'] += 1
        for k, v in d.items():
            agg[name][k] += v
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]['_dur']):
        n = a['_n']
        print(f'{name[:100]}: {int(n)} launches, {a["_dur"] / n / 1e3:.1f} us')
        wc = a.get('SQ_WAVE_CYCLES'); gr = a.get('GRBM_GUI_ACTIVE')
        for k in sorted(a):
            if k.startswith('_'):
                continue
            extra = ''
            if wc and k != 'SQ_WAVE_CYCLES':
                extra += f'   {a[k] / wc:8.4f} per SQ_WAVE_CYCLES'
            if gr and k != 'GRBM_GUI_ACTIVE':
                extra += f'   {a[k] / (gr / 8.0):10.2f} per GRBM cycle'
            print(f'    {k:34s} {a[k] / n:16.0f}{extra}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
