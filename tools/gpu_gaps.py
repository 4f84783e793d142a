#!/usr/bin/env python
"""Idle gaps of the GPU in a rocprofv3 kernel trace (rocpd .db): union of kernel intervals over the LAST `frac` of the trace (the
steady-state steps), the largest gaps with the kernels on either side.   gpu_gaps.py results.db [frac=0.5] [top=25]
(frac > 1: the last `frac` MILLISECONDS of the trace instead of a fraction)"""
import re
import sqlite3
import sys


def main(path, frac=0.5, top=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in scols else 'kernel_name'
    rows = sorted(cur.execute(f'select d.start, d.end, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id').fetchall())
    t_end = rows[-1][1]
    t0 = rows[0][0] + (t_end - rows[0][0]) * (1 - frac) if frac <= 1 else t_end - frac * 1e6
    rows = [r for r in rows if r[0] >= t0]
    busy, cs, ce, last, gaps = 0, None, None, None, []
    for a, b, n in rows:
        if ce is None or a > ce:
            if ce is not None:
                busy += ce - cs
                gaps.append((a - ce, last, n, ce - t0))
            cs, ce = a, b
            last = n
        else:
            if b > ce:
                ce, last = b, n
    busy += ce - cs
    span = rows[-1][1] - rows[0][0]
    print(f'window {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms = {100 * busy / span:.1f} %, {len(gaps)} gaps = {sum(g[0] for g in gaps) / 1e6:.1f} ms')
    hist = [(50e3, 0, 0.0), (200e3, 0, 0.0), (1e6, 0, 0.0), (1e12, 0, 0.0)]
    names = ['< 50 us', '50-200 us', '0.2-1 ms', '> 1 ms']
    acc = [[0, 0.0] for _ in hist]
    for g in gaps:
        for i, (lim, _, _) in enumerate(hist):
            if g[0] < lim:
                acc[i][0] += 1; acc[i][1] += g[0]
                break
    print('  '.join(f'{n}: {c} gaps {t / 1e6:.1f} ms' for n, (c, t) in zip(names, acc)))
    sh = lambda n: re.sub(r'\(anonymous namespace\)::|void ', '', n)[:60]
    for g in sorted(gaps, key=lambda g: -g[0])[:top]:
        print(f'{g[0] / 1e3:9.1f} us at +{g[3] / 1e6:8.2f} ms   after {sh(g[1]):60s} before {sh(g[2])}')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else 25)
