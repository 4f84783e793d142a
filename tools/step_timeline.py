#!/usr/bin/env python
"""Kernel timeline of the LAST `ms` milliseconds of a rocprofv3 kernel trace (rocpd .db), one line per dispatch:
    t_start_us  dur_us  stream  grid  kernel
followed by per-stream busy time and the union.   step_timeline.py results.db ms [min_us=0]"""
import re
import sqlite3
import sys


def main(path, ms, min_us=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in scols else 'kernel_name'
    rows = sorted(cur.execute(f'select d.start, d.end, d.stream_id, d.queue_id, d.grid_size_x, d.workgroup_size_x, s.{namecol} '
                              f'from {kd} d join {ks} s on d.kernel_id = s.id').fetchall())
    t_end = rows[-1][1]
    t0 = t_end - ms * 1e6
    rows = [r for r in rows if r[0] >= t0]
    mc = [t for t in tables if t.startswith('rocpd_memory_copy')]
    if mc:                                        # --memory-copy-trace: the D2H / H2D copies as pseudo-dispatches
        for a, b, size, st, q in cur.execute(f'select start, end, size, stream_id, queue_id from {mc[0]} where start >= {t0}').fetchall():
            rows.append((a, b, st, q, 0, 1, f'MEMCPY {size} B'))
        rows.sort()
    sh = lambda n: re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n))[:70]
    per = {}
    for a, b, st, q, gx, wx, n in rows:
        key = (st, q)
        per.setdefault(key, []).append((a, b))
        if (b - a) / 1e3 >= min_us:
            print(f'{(a - t0) / 1e3:10.1f} {(b - a) / 1e3:8.1f}  s{st}q{q} {gx // max(1, wx):6d}  {sh(n)}')
    for key, iv in sorted(per.items()):
        print(f'# stream {key}: {len(iv)} dispatches, {sum(b - a for a, b in iv) / 1e6:.3f} ms of kernel time')
    iv = sorted((a, b) for a, b, *_ in rows)
    busy, cs, ce = 0, None, None
    for a, b in iv:
        if ce is None or a > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    print(f'# union busy {busy / 1e6:.3f} ms of {ms} ms')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
