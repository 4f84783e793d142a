"""Seeded, platform-independent synthetic weights keyed by state-dict entry name.

There is no network in the build/GPU containers, so the pretrained checkpoints the reference downloads
(/root/reference/lvae/models/qarv/zoo.py:92-95, qresvae/zoo.py:51-54) are unavailable.  Benchmarks and
parity tests therefore use random-init weights of the same architecture, generated identically on every
box from (name, shape, seed) with numpy's counter-based Philox generator -- so the 374 MB of qarv_base
weights never has to be shipped.  The default PyTorch init is degenerate for a codec (gamma = 1e-6, zero
biases: /root/reference/lvae/models/common.py:10-13,135 => every symbol is 0), hence the wider
distributions below.
"""
import hashlib

import numpy as np

_SKIP = ('discrete_gaussian.', '_dummy')


def _rng(name, seed):
    h = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    key = int.from_bytes(h[:16], 'little')
    return np.random.Generator(np.random.Philox(key=key))


# 'wide'    : posterior x8 / prior x4 -- symbols span roughly +-20, scale indexes cover all 64 rows, ~11% of the symbols
#             take the coder's bypass escape, ~11 bpp.  Used by the parity tests / golden fixtures (stresses every path).
# 'typical' : posterior x1 / prior x1 -- ~2.5 bpp, no escapes: the coder load of a trained model at lambda=2048
#             (reference Kodak: 2.21 bpp, results/kodak/kodak-qarv_base.json).  Used by bench.py; GPU work is identical.
# third/fourth entries: the same two knobs for QRes-VAE, whose posterior / prior heads are 4-conv VDBlocks (last conv `c4`)
# that attenuate far more than QARV's single convs.
PROFILES = {'wide': (8.0, 4.0, 300.0, 60.0), 'typical': (1.0, 1.0, 40.0, 20.0)}


def seeded_tensor(name, shape, seed=0, profile='wide'):
    """numpy float32 array for one state-dict entry, or None if the entry is a derived buffer."""
    post_scale, prior_scale, post_c4, prior_c4 = PROFILES[profile]
    if any(s in name for s in _SKIP):
        return None
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'gamma':
        a = g.uniform(0.05, 0.25, size=shape)
    elif name == 'bias' or name == 'decoder.bias':
        a = g.normal(0.0, 0.5, size=shape)
    elif leaf == 'bias':
        a = g.normal(0.0, 0.05, size=shape)
    elif leaf == 'weight' and len(shape) == 1:           # LayerNorm affine weight
        a = 1.0 + 0.1 * g.normal(0.0, 1.0, size=shape)
    elif leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        a = g.uniform(-b, b, size=shape)
        if name.endswith('.posterior.weight'):
            a *= post_scale
        elif name.endswith('.prior.weight'):
            a *= prior_scale
        elif '.posterior.c4.' in name:
            a *= post_c4
        elif '.prior.c4.' in name:
            a *= prior_c4
    else:
        a = g.normal(0.0, 0.05, size=shape)
    return a.astype(np.float32)


def seeded_state_dict(named_shapes, seed=0, profile='wide'):
    """named_shapes: iterable of (name, shape).  Returns {name: np.float32 array} for parameter entries."""
    out = {}
    for name, shape in named_shapes:
        a = seeded_tensor(name, shape, seed, profile)
        if a is not None:
            out[name] = a
    return out


def synthetic_image_u8(h, w, seed=0, kind='natural'):
    """Seeded uint8 HxWx3 test image (SURVEY.md 8(d) 'Synthetic inputs').

    'natural': 8x bilinearly-upsampled uniform noise + N(0, 4) pixel noise, clipped;
    'noise'  : pure uniform noise (high-entropy worst case for the coder).
    """
    g = _rng(f'image:{kind}:{h}x{w}', seed)
    if kind == 'noise':
        return g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    gh, gw = h // 8 + 2, w // 8 + 2
    coarse = g.uniform(0.0, 255.0, size=(gh, gw, 3))
    ys = (np.arange(h) + 0.5) / 8.0
    xs = (np.arange(w) + 0.5) / 8.0
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    c00 = coarse[y0][:, x0]
    c01 = coarse[y0][:, x0 + 1]
    c10 = coarse[y0 + 1][:, x0]
    c11 = coarse[y0 + 1][:, x0 + 1]
    img = (c00 * (1 - fy) * (1 - fx) + c01 * (1 - fy) * fx + c10 * fy * (1 - fx) + c11 * fy * fx)
    img = img + g.normal(0.0, 4.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
