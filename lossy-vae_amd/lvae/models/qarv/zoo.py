"""QARV model zoo (reference: lvae/models/qarv/zoo.py:9-99).

`qarv_base`: 93.4 M parameters, max stride 64, lambda range (16, 2048), 9 latent blocks with
z = 32 | 32,32 | 96,96,96 | 8,8,8 at strides 64 | 32 | 16 | 8.  The network is described as two rows-of-stages
tables instead of module lists; `get_model('qarv_base', pretrained=path)` loads the reference's checkpoints
(`torch.load(path)['model']`, identical key names).
"""
import torch

from ..registry import register_model
from . import model as qarv

# (stride, channels, [(dw kernel, count), ...], tap key after how many blocks)  -- zoo.py:35-60
_ENC_STAGES = [
    (4, 192, [(7, 7)], None),
    (2, 384, [(7, 7)], ('enc_s8', 6)),
    (2, 512, [(5, 6), (7, 1)], ('enc_s16', 6)),
    (2, 512, [(3, 4), (7, 1)], ('enc_s32', 4)),
    (2, 512, [(1, 4)], ('enc_s64', 4)),
]
# (width, z, n_latent, enc_key, enc_width, kernel, mlp_ratio, next width, upsample rate)  -- zoo.py:62-88
_DEC_STAGES = [
    (512, 32, 1, 'enc_s64', 512, 1, 4, 512, 2),
    (512, 32, 2, 'enc_s32', 512, 3, 3, 384, 2),
    (384, 96, 3, 'enc_s16', 512, 5, 2, 256, 2),
    (256, 8, 3, 'enc_s8', 384, 7, 1.75, 128, 2),
]


def _qarv_base_blocks():
    enc, cin = [], 3
    for stride, ch, groups, tap in _ENC_STAGES:
        enc.append(qarv.DownParams(cin, ch, stride))
        n = 0
        for k, cnt in groups:
            for _ in range(cnt):
                if tap and n == tap[1]:
                    enc.append(qarv._Marker('key', tap[0]))
                enc.append(qarv.CNXParams(ch, kernel_size=k))
                n += 1
        if tap and n == tap[1]:
            enc.append(qarv._Marker('key', tap[0]))
        cin = ch
    dec = []
    for si, (w, z, nlat, key, ew, k, mlp, nxt, rate) in enumerate(_DEC_STAGES):
        if si > 0:
            dec.append(qarv.CNXParams(w, kernel_size=k, mlp_ratio=mlp))
        dec += [qarv.VRLVParams(w, z, enc_key=key, enc_width=ew, kernel_size=k, mlp_ratio=mlp) for _ in range(nlat)]
        if si == len(_DEC_STAGES) - 1:
            dec.append(qarv._Marker('stop'))          # CompresionStopFlag, zoo.py:82
        dec.append(qarv.CNXParams(w, kernel_size=k, mlp_ratio=mlp))
        dec.append(qarv.UpParams(w, nxt, rate))
    dec += [qarv.CNXParams(128, kernel_size=7, mlp_ratio=1.5) for _ in range(8)]
    dec.append(qarv.UpParams(128, 3, 4))
    return enc, dec


@register_model
def qarv_base(lmb_range=(16, 2048), pretrained=False):
    cfg = dict()
    cfg['im_shift'] = -0.4546259594901961      # mean and std computed on imagenet (zoo.py:14-15)
    cfg['im_scale'] = 3.67572653978347
    cfg['max_stride'] = 64
    cfg['lmb_range'] = (float(lmb_range[0]), float(lmb_range[1]))
    cfg['lmb_embed_dim'] = (256, 256)
    cfg['sin_period'] = 64
    cfg['enc_blocks'], cfg['dec_blocks'] = _qarv_base_blocks()
    model = qarv.VariableRateLossyVAE(cfg)
    if pretrained is True:
        from torch.hub import load_state_dict_from_url
        url = 'https://huggingface.co/duanzh0/my-model-weights/resolve/main/qarv_base-2022-dec-12.pt'
        model.load_state_dict(load_state_dict_from_url(url)['model'])
    elif pretrained:    # str or Path
        model.load_state_dict(torch.load(pretrained)['model'])
    return model
