#!/usr/bin/env python
"""MFMA utilisation of the dominant kernel family at the bench workload, from one rocprofv3 PMC pass (--kernel-trace --pmc
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY):   pmc_mfma_util.py pass.db
Per kernel: launches, mean duration, MFMA-busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
(SQ_VALU_MFMA_BUSY_CYCLES counts pipe-busy cycles summed over the chip's SIMDs: 32 per v_mfma_f32_32x32x16_f16; GRBM_GUI_ACTIVE is summed over
the 8 XCDs: MI355X_MICROARCH.md, rocprofv3 section), effective clock = GRBM_GUI_ACTIVE / 8 / duration."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    T = {re.sub(r'_[0-9a-f]{8}_.*$', '', r[0]): r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    q = f'''select s.display_name, p.name, e.value, d.id, d.end - d.start from {T['rocpd_pmc_event']} e
            join {T['rocpd_info_pmc']} p on e.pmc_id = p.id
            join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id
            join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id'''
    per = defaultdict(lambda: defaultdict(float))
    for name, pmc, val, did, dur in cur.execute(q):
        k = (re.sub(r'\s+', ' ', name.replace('(anonymous namespace)::', '')), did)
        per[k][pmc] += val
        per[k]['dur'] = dur
    agg = defaultdict(lambda: defaultdict(float))
    for (name, did), d in per.items():
        a = agg[name]
        a['n'] += 1
        for k, v in d.items():
            a[k] += v
    fam = defaultdict(float)
    rows = []
    for name, a in agg.items():
        if not a.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            continue
        simd_cycles = 1024.0 * a['GRBM_GUI_ACTIVE'] / 8.0
        rows.append((a['dur'], name, a))
        if re.search(r'gemm_h2p_kernel<|gemm_h2_kernel<\d, (true|false), 0>|mlp_h2c_kernel<|mlp_sk_kernel<|mlp_h2f_kernel', name):
            for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'dur', 'n', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_INSTS_MFMA'):
                fam[k] += a.get(k, 0.0)
    rows.sort(key=lambda r: -r[0])
    print(f'{"kernel":86s} {"launches":>8s} {"us/launch":>10s} {"MFMA-busy %":>12s} {"clock GHz":>10s} {"waves parked %":>15s}')
    for dur, name, a in rows[:24]:
        simd_cycles = 1024.0 * a['GRBM_GUI_ACTIVE'] / 8.0
        park = 100.0 * a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES'] if a.get('SQ_WAVE_CYCLES') else float('nan')
        print(f'{re.sub(r"[(].*$", "", name)[:86]:86s} {int(a["n"]):8d} {a["dur"] / a["n"] / 1e3:10.1f} {100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles:12.1f} '
              f'{a["GRBM_GUI_ACTIVE"] / 8.0 / a["dur"]:10.2f} {park:15.1f}')
    if fam['n']:
        simd_cycles = 1024.0 * fam['GRBM_GUI_ACTIVE'] / 8.0
        print(f'# dominant family (gemm_h2p + gemm_h2<*, *, 0> + fused MLP kernels): {int(fam["n"])} launches, {fam["dur"] / fam["n"] / 1e3:.1f} us per launch, '
              f'MFMA-busy {100.0 * fam["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles:.1f} % of the SIMD cycles of these launches '
              f'({fam["SQ_INSTS_MFMA"] / fam["n"]:.0f} MFMA wave-instructions per launch), effective clock {fam["GRBM_GUI_ACTIVE"] / 8.0 / fam["dur"]:.2f} GHz, '
              f'waves parked (s_waitcnt / barrier) {100.0 * fam["SQ_WAIT_ANY"] / fam["SQ_WAVE_CYCLES"]:.1f} % of their cycles')


if __name__ == '__main__':
    main(sys.argv[1])
