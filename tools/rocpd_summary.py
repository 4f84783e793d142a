#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, % of GPU kernel time) from a rocprofv3 rocpd .db
(`rocprofv3 --kernel-trace --stats`), equivalent to its kernel_stats table.  Usage: rocpd_summary.py x_results.db"""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in cur.execute(f'pragma table_info({kd})')]
    scols = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
    namecol = 'display_name' if 'display_name' in scols else 'kernel_name'
    rows = cur.execute(f'select s.{namecol}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id').fetchall()
    iv = sorted(cur.execute(f'select d.start, d.end from {kd} d').fetchall())
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for a, b in iv:
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(a - cur_e)
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0] if iv else 1
    big = sorted(gaps)[-20:]
    print(f'# GPU busy (union of kernel intervals) {busy / 1e6:.1f} ms of a {span / 1e6:.1f} ms span = {100 * busy / span:.1f} %; '
          f'{len(gaps)} idle gaps, {sum(gaps) / 1e6:.1f} ms in total, {sum(g for g in gaps if g > 50000) / 1e6:.1f} ms of it in gaps > 50 us')
    agg = {}
    for name, dur in rows:
        name = re.sub(r'\s+', ' ', name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    fam = [0, 0]
    for name, a in agg.items():
        if (re.search(r'gemm(_x3|_bf16)?_kernel', name) and name.rstrip().endswith(', 0>(lvae_gemm_desc, int, int)')) or \
                re.search(r'gemm_x3k16_kernel<\d, (true|false), 0>|gemm_x3w8_kernel|gemm_lp_kernel<\d, 0, |gemm_h2_kernel<\d, (true|false), 0>|gemm_h2p_kernel<|gemm_q8_kernel<|mlp_h2c_kernel<|mlp_sk_kernel<', name):
            fam[0] += a[0]; fam[1] += a[1]
    tot = sum(a[1] for a in agg.values())
    print(f'# {path}: {len(rows)} dispatches, total kernel time {tot / 1e6:.3f} ms')
    if fam[0]:
        print(f'# family gemm[_x3|_bf16]_kernel<Cfg<*>, 0> + gemm_x3k16/x3w8/lp/h2/h2p/q8_kernel (PLAIN, all tile configs) + mlp_h2c_kernel / mlp_sk_kernel (fused fc1 -> GELU -> fc2): {fam[0]} calls, avg {fam[1] / fam[0] / 1e3:.2f} us, '
              f'total {fam[1] / 1e6:.3f} ms  <- compare with bench.py roofline.avg_launch_us')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"pct":>6}  kernel')
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:6.2f}  {name[:150]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
