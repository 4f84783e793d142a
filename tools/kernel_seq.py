#!/usr/bin/env python
"""Kernel sequence of the last `ms` milliseconds of a rocprofv3 kernel trace (rocpd .db): start offset, gap since the previous kernel's
end, duration, short name -- the launch-by-launch picture of one latency-bound chain (a B = 1 decode is one stream).
    kernel_seq.py results.db ms [name-width]"""
import re
import sqlite3
import sys


def main(path, ms, width=70):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in scols else 'kernel_name'
    rows = sorted(cur.execute(f'select d.start, d.end, s.{namecol}, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id').fetchall())
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - ms * 1e6]
    t0, prev = rows[0][0], rows[0][0]
    busy = 0
    for a, b, n, gx, wx in rows:
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*$', '', n)
        print(f'{(a - t0) / 1e3:9.1f} us  gap {(a - prev) / 1e3:6.1f}  dur {(b - a) / 1e3:6.1f}  wgs {gx // max(1, wx):6d}  {n[:width]}')
        busy += b - a
        prev = max(prev, b)
    print(f'# {len(rows)} kernels, busy {busy / 1e3:.1f} us of {(rows[-1][1] - t0) / 1e3:.1f} us')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 70)
