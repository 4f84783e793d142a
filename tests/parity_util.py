"""Shared by the -m gpu parity tests: prove that every index / symbol flip between the HIP path and the reference (goldens or the
live CPU oracle) is a GUARD-BAND event -- a value that sits on a decision threshold to within fp32 rounding noise on BOTH sides --
and that nothing else differs.

The comparison is TEACHER-FORCED: the HIP encoder runs block by block and hands each block's successors the REFERENCE's quantised
latent (`encode_trace(..., full=True, force_z=...)`), so every block sees the reference's inputs up to rounding noise and a flip is
never the cascade of an earlier flip.  Then, per latent block:

  * closeness: max |pm - pm_ref|, |qm - qm_ref| <= VAL_ATOL + VAL_RTOL * |ref| and max |ln sigma - ln sigma_ref| <= LNS_TOL for ALL
    elements (not only the flipped ones) -- a kernel bug that moves a few values per million fails here;
  * index flips: |i - i_ref| == 1 and both sigmas within IDX_BAND (relative) of the table threshold between the two indexes;
  * symbol flips: |s - s_ref| == 1 and both (qm - pm) within SYM_BAND of the half-integer between the two symbols.

The bands are 3-10x the rounding noise measured on the MI355X at 512x768 (2.1 M teacher-forced elements of qres34m and of a batch-of-8
qarv_base encode, 'wide' seeded weights: every element within 3.4e-5 / 1.3e-5 in ln sigma; the 22 flips within 1.3e-5 of their
half-integer / 3.0e-6 relative of their table threshold) and 10^3..10^4 times smaller than the quantisation step they guard.
"""
import math

import numpy as np

VAL_ATOL, VAL_RTOL = 1e-4, 2e-5       # pm, qm (values up to a few hundred with the 'wide' seeded weights); measured max 3.4e-5
LNS_TOL = 5e-5                        # ln sigma; measured max 1.3e-5
IDX_BAND = 3e-5                       # |sigma / threshold - 1| of a flipped scale index, on both sides; measured worst 3.0e-6
SYM_BAND = 1e-4                       # | |qm - pm| - (k + 1/2) | of a flipped symbol, on both sides (+ VAL_RTOL * |qm - pm|); measured worst 1.3e-5


def sigma_from_lv(lv, bound):
    """exp(softplus(lv + 2.3) - 2.3) clamped below (qarv/model.py:51-53, LowerBound) in float64."""
    lv = np.asarray(lv, dtype=np.float64)
    xs = lv + np.float64(np.float32(2.3))
    sp = np.where(xs > 20.0, xs, np.log1p(np.exp(np.minimum(xs, 20.0))))
    return np.maximum(np.exp(sp - np.float64(np.float32(2.3))), bound)


def check_blocks(case, gpu_blocks, ref_blocks, table, bound, rows=None):
    """gpu_blocks: encode_trace(full=True, force_z=ref latents) of the product model, arrays (B, z, hw).
    ref_blocks: per block dict(pm, pv (sigma before the lower bound), qm, indexes, symbols) with arrays reshapeable to (Bsel, z, hw);
    rows: batch rows of gpu_blocks that ref_blocks describe (default: all).
    Returns dict(sym_flips, idx_flips, n, max_dval, max_dlns, worst_sym_margin, worst_idx_margin); raises AssertionError on the first
    element outside its band."""
    table = np.asarray(table, dtype=np.float64)
    st = dict(sym_flips=0, idx_flips=0, n=0, max_dval=0.0, max_dlns=0.0, worst_sym_margin=0.0, worst_idx_margin=0.0)
    for bi, (a, r) in enumerate(zip(gpu_blocks, ref_blocks)):
        sel = slice(None) if rows is None else list(rows)
        shp = a['symbols'][sel].shape
        g = {k: np.asarray(a[k][sel]) for k in ('symbols', 'indexes', 'pm', 'lv', 'qm')}
        ref = {k: np.asarray(r[k].numpy() if hasattr(r[k], 'numpy') else r[k]).reshape(shp) for k in ('symbols', 'indexes', 'pm', 'pv', 'qm')}
        st['n'] += g['symbols'].size
        # ---- closeness of every element
        for k in ('pm', 'qm'):
            d = np.abs(g[k].astype(np.float64) - ref[k])
            tol = VAL_ATOL + VAL_RTOL * np.abs(ref[k])
            st['max_dval'] = max(st['max_dval'], float(d.max()))
            bad = d > tol
            assert not bad.any(), f'{case}: block {bi} {k}: {int(bad.sum())} elements beyond rounding noise, worst {float(d.max()):.3e} (tol {float(tol[np.unravel_index(d.argmax(), d.shape)]):.3e})'
        sg = sigma_from_lv(g['lv'], bound)
        sr = np.maximum(ref['pv'].astype(np.float64), bound)
        dl = np.abs(np.log(sg) - np.log(sr))
        st['max_dlns'] = max(st['max_dlns'], float(dl.max()))
        assert float(dl.max()) <= LNS_TOL, f'{case}: block {bi} ln sigma differs by {float(dl.max()):.3e}'
        # ---- index flips: adjacent indexes, both sigmas on the threshold between them
        fi = np.nonzero(g['indexes'] != ref['indexes'])
        st['idx_flips'] += len(fi[0])
        for pos in zip(*fi):
            ig, ir = int(g['indexes'][pos]), int(ref['indexes'][pos])
            assert abs(ig - ir) == 1, f'{case}: block {bi} index {ig} vs {ir} at {pos}: not adjacent'
            thr = table[min(ig, ir)]              # index = #{i < n-1 : table[i] < sigma}: the two indexes are separated by table[min]
            m = max(abs(sg[pos] / thr - 1), abs(sr[pos] / thr - 1))
            st['worst_idx_margin'] = max(st['worst_idx_margin'], float(m))
            assert m <= IDX_BAND, f'{case}: block {bi} index flip at {pos}: sigma {sg[pos]:.9g} / {sr[pos]:.9g} vs threshold {thr:.9g} (margin {m:.3e})'
        # ---- symbol flips: adjacent integers, both pre-round values on the half-integer between them
        fs = np.nonzero(g['symbols'] != ref['symbols'])
        st['sym_flips'] += len(fs[0])
        for pos in zip(*fs):
            s_g, s_r = int(g['symbols'][pos]), int(ref['symbols'][pos])
            assert abs(s_g - s_r) == 1, f'{case}: block {bi} symbol {s_g} vs {s_r} at {pos}: not adjacent'
            half = (s_g + s_r) / 2.0
            vg = float(np.float32(g['qm'][pos]) - np.float32(g['pm'][pos]))
            vr = float(np.float32(ref['qm'][pos]) - np.float32(ref['pm'][pos]))
            m = max(abs(vg - half), abs(vr - half))
            st['worst_sym_margin'] = max(st['worst_sym_margin'], m)
            assert m <= SYM_BAND + VAL_RTOL * abs(half), f'{case}: block {bi} symbol flip at {pos}: qm-pm {vg:.7f} / {vr:.7f} vs {half} (margin {m:.3e})'
    return st


def describe(st):
    return (f"teacher-forced: {st['sym_flips']} symbol + {st['idx_flips']} index flips of {st['n']}, ALL inside the guard band "
            f"(worst margins {st['worst_sym_margin']:.1e} / {st['worst_idx_margin']:.1e}); every element: max|d pm,qm| {st['max_dval']:.1e}, "
            f"max|d ln sigma| {st['max_dlns']:.1e}")
