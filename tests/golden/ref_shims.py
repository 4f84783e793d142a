"""tests/golden/ref_shims.py -- build-container-only helper.

Lets the reference's own Python (/root/reference, read-only) be imported in THIS container, where
`compressai`, `timm` and `torchvision` are not installed, by registering small stand-in modules for
the handful of third-party symbols the hot path touches.  The stand-ins
are this repo's own code: trivial pieces restated here, CompressAI semantics from
oracle/compressai_semantics.py.  Nothing under /root/reference is copied.

Used ONLY by tests/golden/make_golden.py to generate the committed fixtures; it never runs on the GPU
box (where /root/reference does not exist).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = '/root/reference'


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from oracle import compressai_semantics as cs

    # ---- torchvision
    tv = _mod('torchvision')
    tvu = _mod('torchvision.utils')
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    tvt = _mod('torchvision.transforms')
    tvf = _mod('torchvision.transforms.functional')

    def to_tensor(pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous()
        return t.to(dtype=torch.float32).div(255) if t.dtype == torch.uint8 else t.float()

    def pad(img, padding, padding_mode='edge'):
        from PIL import Image
        left, top, right, bottom = padding
        a = np.asarray(img)
        pw = ((top, bottom), (left, right)) + (((0, 0),) if a.ndim == 3 else ())
        return Image.fromarray(np.pad(a, pw, mode='edge'))

    tvf.to_tensor, tvf.pad = to_tensor, pad
    tvt.functional = tvf
    tv.transforms = tvt

    # ---- timm
    timm = _mod('timm')
    tu = _mod('timm.utils')

    class AverageMeter:
        def __init__(self):
            self.sum, self.count, self.avg, self.val = 0.0, 0, 0.0, 0.0

        def update(self, val, n=1):
            self.val = val
            self.sum += val * n
            self.count += n
            self.avg = self.sum / self.count

    tu.AverageMeter = AverageMeter
    timm.utils = tu
    tl = _mod('timm.layers')
    tlm = _mod('timm.layers.mlp')

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, **kw):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features or in_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    tlm.Mlp = Mlp
    tl.mlp = tlm
    tl.Mlp = Mlp
    timm.layers = tl
    tm = _mod('timm.models')
    tmc = _mod('timm.models.convnext')

    class ConvNeXtBlock(nn.Module):
        def __init__(self, dim, kernel_size=7, mlp_ratio=4, ls_init_value=1e-6, **kw):
            super().__init__()
            self.use_conv_mlp = False
            self.conv_dw = nn.Conv2d(dim, dim, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, groups=dim)
            self.norm = nn.LayerNorm(dim, eps=1e-6)
            self.mlp = Mlp(dim, int(mlp_ratio * dim))
            self.gamma = nn.Parameter(ls_init_value * torch.ones(dim)) if ls_init_value is not None else None
            self.drop_path = nn.Identity()

    tmc.ConvNeXtBlock = ConvNeXtBlock
    tm.convnext = tmc
    timm.models = tm

    # ---- compressai
    ca = _mod('compressai')
    cao = _mod('compressai.ops')
    cao.LowerBound = cs.LowerBound
    cae = _mod('compressai.entropy_models')
    cae.GaussianConditional = cs.GaussianConditional
    cae.EntropyModel = cs.EntropyModel
    ca.ops, ca.entropy_models = cao, cae

    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True  # /root/reference is read-only
