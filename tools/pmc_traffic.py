#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each its own run with --kernel-trace only).
    pmc_traffic.py fetch.db write.db [out.json]
Units: FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes
of a wide coalesced streaming read -> 'x2' column; WRITE_SIZE is reported as is (round 6: ratio 1.000 against known byte counts in this library's store patterns, profiles/r06_write_size_calibration.txt).  'MB' in the output are MiB (the counters are KiB)."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    T = {re.sub(r'_[0-9a-f]{8}_.*$', '', r[0]): r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    q = f'''select s.display_name, p.name, e.value, d.id from {T['rocpd_pmc_event']} e
            join {T['rocpd_info_pmc']} p on e.pmc_id = p.id
            join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id
            join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id'''
    per = defaultdict(float)
    for name, pmc, val, did in cur.execute(q):
        if pmc == counter:
            per[(re.sub(r'\s+', ' ', name), did)] += val
    agg = defaultdict(lambda: [0, 0.0])
    for (name, did), v in per.items():
        a = agg[name]
        a[0] += 1
        a[1] += v
    return agg


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)[:78]


def main(fetch_db, write_db, out_json=None):
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    rows = []
    for name in f:
        n, kib = f[name]
        wn, wkib = w.get(name, [0, 0.0])
        rows.append((name, n, kib / n / 1024.0, (wkib / wn / 1024.0) if wn else 0.0))
    rows.sort(key=lambda r: -r[1] * (2 * r[2] + r[3]))
    print(f'{"kernel":78s} {"launches":>8s} {"fetch MB/launch":>16s} {"x2":>10s} {"write MB/launch":>16s}')
    for name, n, fm, wm in rows[:40]:
        print(f'{short(name):78s} {n:8d} {fm:16.2f} {2 * fm:10.2f} {wm:16.2f}')
    fam = [r for r in rows if (re.search(r'gemm(_x3|_bf16)?_kernel', r[0]) and r[0].rstrip().endswith(', 0>(lvae_gemm_desc, int, int)'))
           or re.search(r'gemm_x3k16_kernel<\d, (true|false), 0>|gemm_x3w8_kernel|gemm_lp_kernel<\d, 0, |gemm_h2_kernel<\d, (true|false), 0>|gemm_h2p_kernel<|gemm_q8_kernel<|mlp_h2c_kernel<|mlp_sk_kernel<|mlp_h2f_kernel', r[0])]
    n = sum(r[1] for r in fam)
    if n:
        fm = sum(r[1] * r[2] for r in fam) / n
        wm = sum(r[1] * r[3] for r in fam) / n
        print(f'# PLAIN GEMM family: {n} launches, fetch {fm:.2f} MB/launch (x2 = {2 * fm:.2f}), write {wm:.2f} MB/launch, '
              f'corrected total {2 * fm + wm:.2f} MB/launch')
        if out_json:
            json.dump({'family': 'PLAIN GEMM launches (gemm_h2p/h2/q8/x3k16/x3w8/x3/lp/gemm kernels, AMODE 0)', 'launches': n,
                       'fetch_mb_per_launch_raw': round(fm, 3), 'fetch_mb_per_launch_x2': round(2 * fm, 3),
                       'write_mb_per_launch': round(wm, 3), 'hbm_mb_per_launch_corrected': round(2 * fm + wm, 3),
                       'correction': 'FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is (calibrated: ratio 1.000, profiles/r06_write_size_calibration.txt); units MiB'},
                      open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:4])
