"""CPU study (build container): how many symbol / index flips against the reference goldens does each OPERAND SPLIT of the
channel-mixing GEMMs cost by itself?  The oracle's F.linear / F.conv2d (groups == 1) are replaced by an emulation that splits both
operands as the kernels do, forms the kept cross terms exactly (fp64) and rounds the sum to fp32 once -- i.e. everything except the
fp32 accumulation order of the MFMA pipe, which is common to all splits.

  exact     : no split (fp64 products of the fp32 operands): the floor set by the reference's own fp32 accumulation noise
  bf16x3    : x = hi + mid + lo (bf16 each), 6 cross terms (the round-1/2 default arithmetic)
  f16x2_3   : x = hi + lo (fp16 each, lo kept scaled), terms hh + hl + lh   (3 MFMAs)
  f16x2_4   : the same + ll                                                  (4 MFMAs)

usage: python tools/split_error_study.py [64x64|128x192] [modes...]
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'lossy-vae_amd'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import seeded_init                                            # noqa: E402
from oracle import qarv_oracle                                # noqa: E402


def split_bf16(x, n):
    parts, r = [], x.float()
    for _ in range(n):
        p = r.to(torch.bfloat16).float()
        parts.append(p.double())
        r = r - p
    return parts


def split_f16(x):
    x = x.float()
    hi = x.to(torch.float16).float()
    lo = ((x - hi) * 2048.0).to(torch.float16).float() / 2048.0
    return [hi.double(), lo.double()]


def make_F(mode):
    def terms(a, w):
        if mode == 'exact':
            return [(a.double(), w.double())]
        if mode == 'bf16x3':
            A, W = split_bf16(a, 3), split_bf16(w, 3)
            return [(A[i], W[j]) for i in range(3) for j in range(3) if i + j <= 2]
        A, W = split_f16(a), split_f16(w)
        t = [(A[0], W[0]), (A[0], W[1]), (A[1], W[0])]
        if mode == 'f16x2_4':
            t.append((A[1], W[1]))
        return t

    def linear(x, w, b=None):
        if x.numel() <= 4096:                     # lambda embedding / AdaLN GEMVs: plain fp32 kernels in the product too
            return F.linear(x, w, b)
        acc = sum(F.linear(a, ww) for a, ww in terms(x, w)).float()
        return acc + b if b is not None else acc

    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 or w.shape[1] == 3:        # depthwise conv / the 4x4 stem: fp32 FMA kernels, not GEMMs
            return F.conv2d(x, w, b, stride, padding, dilation, groups)
        acc = sum(F.conv2d(a, ww, None, stride, padding) for a, ww in terms(x, w)).float()
        return acc + b.view(1, -1, 1, 1) if b is not None else acc

    ns = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith('_')})
    ns.linear, ns.conv2d = linear, conv2d
    return ns


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else '64x64'
    modes = sys.argv[2:] or ['exact', 'bf16x3', 'f16x2_3', 'f16x2_4']
    g = np.load(os.path.join(REPO, 'tests', 'golden', f'qarv_base_{tag}.npz'))
    h, w = g['hw'].tolist()
    seed = {'64x64': 0, '128x192': 1}[tag]
    u8 = seeded_init.synthetic_image_u8(h, w, seed)
    im = torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0)
    arch = qarv_oracle.qarv_base_arch()
    sd = seeded_init.seeded_state_dict(qarv_oracle.qarv_param_shapes(arch), seed=0)
    m = qarv_oracle.QarvOracle(sd)
    m.compress_mode()
    real_F = qarv_oracle.F
    exact = {}
    for mode in modes:
        qarv_oracle.F = make_F(mode)
        tot = [0, 0, 0]
        worst = 0.0
        se, sr, n = 0.0, 0.0, 0            # squared deviation of qm / pm from the 'exact' run, and of the golden from it
        for lmb in g['lmbs'].tolist():
            key = f'lmb{int(lmb)}'
            tr = m.encode_trace(im, lmb, code=False)
            for bi, blk in enumerate(tr['blocks']):
                tot[0] += int((blk['symbols'].numpy() != g[f'{key}.b{bi}.symbols']).sum())
                tot[1] += int((blk['indexes'].numpy() != g[f'{key}.b{bi}.indexes']).sum())
                tot[2] += blk['symbols'].numel()
                for nm in ('pm', 'qm'):
                    if mode == 'exact':
                        exact[(key, bi, nm)] = blk[nm].double().numpy()
                    elif (key, bi, nm) in exact and f'{key}.b{bi}.{nm}' in g:
                        e = exact[(key, bi, nm)]
                        se += float(((blk[nm].double().numpy() - e) ** 2).sum()); n += e.size
                        sr += float(((g[f'{key}.b{bi}.{nm}'].astype(np.float64) - e) ** 2).sum())
                worst = max(worst, float(np.abs(blk['qm'].numpy() - g[f'{key}.b{bi}.qm']).max()) if f'{key}.b{bi}.qm' in g else 0.0,
                            float(np.abs(blk['pm'].numpy() - g[f'{key}.b{bi}.pm']).max()))
        print(f'{tag} {mode:8s}: symbol flips {tot[0]}  index flips {tot[1]}  of {tot[2]}   max|d pm/qm| {worst:.3e}'
              + (f'   rms dev from exact: this {np.sqrt(se / n):.3e}, reference fp32 {np.sqrt(sr / n):.3e}' if n else ''), flush=True)
    qarv_oracle.F = real_F


if __name__ == '__main__':
    torch.set_num_threads(16)
    main()
