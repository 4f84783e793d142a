"""python tools/r6_write_calib_summary.py known.json write.db fetch.db  -> counter / known bytes per kernel of tools/r6_write_calib.py"""
import json, re, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_traffic import per_kernel
known = json.load(open(sys.argv[1]))
w, f = per_kernel(sys.argv[2], 'WRITE_SIZE'), per_kernel(sys.argv[3], 'FETCH_SIZE')
pats = {'fill': r'FillFunctor', 'fc1': r'gemm_h2p_kernel', 'fc2': r'gemm_h2p_kernel', 'mlp_h2c': r'mlp_h2c_kernel<192, 384', 'dwconv': r'dwconv_ln_cl_kernel<7'}
print(f'{"kernel (launches)":66s} {"known write MB":>15s} {"WRITE_SIZE MB":>14s} {"ratio":>7s} | {"known read MB":>14s} {"FETCH_SIZE x2 MB":>17s} {"ratio":>7s}')
for name, kb in known.items():
    key = name.split(' ')[0].split('_kernel')[0]
    pat = [v for k, v in pats.items() if name.startswith(k)][0]
    cands_w = [(n, v) for n, v in w.items() if re.search(pat, n)]
    cands_f = [(n, v) for n, v in f.items() if re.search(pat, n)]
    # fc1 / fc2 are two instantiations / launch groups of the same kernel family: tell them apart by the written bytes per launch
    def pick(cands):
        if not cands: return None
        if name.startswith('fc'):
            allc = sorted(cands, key=lambda nv: nv[1][1] / max(1, nv[1][0]))
            if len(allc) == 1: return allc[0]
            return allc[-1] if name.startswith('fc1') else allc[0]
        return max(cands, key=lambda nv: nv[1][0])
    cw, cf = pick(cands_w), pick(cands_f)
    wm = cw[1][1] / cw[1][0] / 1024.0 if cw else float('nan')
    fm = 2 * cf[1][1] / cf[1][0] / 1024.0 if cf else float('nan')
    print(f'{(name + " (" + str(cw[1][0] if cw else 0) + ")")[:66]:66s} {kb["write"] / 1e6 * 1e6 / 2**20:15.2f} {wm:14.2f} {wm / (kb["write"] / 2**20):7.3f} | '
          f'{kb["read"] / 2**20:14.2f} {fm:17.2f} {(fm / (kb["read"] / 2**20)) if kb["read"] else float("nan"):7.3f}')
