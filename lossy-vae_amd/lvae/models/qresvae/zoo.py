"""QRes-VAE model zoo (reference: lvae/models/qresvae/zoo.py:9-60): `qres34m`, 34.0 M parameters, one model per lambda
in {16, 32, ..., 2048}, 12 latent blocks with z = 16 | 14,14 | 12,12,12 | 10,10,10 | 8,8,8 at strides 64 ... 4."""
import torch

from ..registry import register_model
from ..qarv.model import UpParams
from . import model as qres

_LAMBDAS = {16, 32, 64, 128, 256, 512, 1024, 2048}


@register_model
def qres34m(lmb=32, pretrained=False):
    ch = 96
    enc_nums, dec_nums, z_dims = [6, 6, 6, 4, 2], [1, 2, 3, 3, 3], [16, 14, 12, 10, 8]
    enc_k, dec_k = [7, 7, 5, 3, 1], [1, 3, 5, 7, 7]
    enc_w = [ch * 2, ch * 4, ch * 4, ch * 4, ch * 4]
    dec_w = [ch * 4, ch * 4, ch * 4, ch * 4, ch * 2]
    enc = [qres.StemParams(3, enc_w[0], 4)]
    for lvl in range(5):
        enc += [qres.MyCNXParams(enc_w[lvl], kernel_size=enc_k[lvl]) for _ in range(enc_nums[lvl])]
        if lvl < 4:
            enc.append(qres.MyCNXDownParams(enc_w[lvl], enc_w[lvl + 1]))
    dec = []
    for lvl in range(5):
        dec += [qres.QLBParams(dec_w[lvl], z_dims[lvl], kernel_size=dec_k[lvl]) for _ in range(dec_nums[lvl])]
        dec.append(UpParams(dec_w[lvl], dec_w[lvl + 1], 2) if lvl < 4 else UpParams(dec_w[lvl], 3, 4))
    cfg = dict(enc_blocks=enc, dec_blocks=dec, im_shift=-0.4546259594901961, im_scale=3.67572653978347, max_stride=64)
    model = qres.HierarchicalVAE(cfg)
    model.mse_lmb = float(lmb)          # MSEOutputNet(mse_lmb=lmb) carries no parameters (qresvae/model.py:97-117)
    if (pretrained is True) and (lmb in _LAMBDAS):
        from torch.hub import load_state_dict_from_url
        url = f'https://huggingface.co/duanzh0/my-model-weights/resolve/main/qres34m/qres34m-lmb{lmb}.pt'
        model.load_state_dict(load_state_dict_from_url(url)['model'])
    elif isinstance(pretrained, str):
        model.load_state_dict(torch.load(pretrained)['model'])
    else:
        assert pretrained is False, f'Invalid {pretrained=} and {lmb=}'
    return model


@register_model
def qres34m_lossless(pretrained=False):
    """(reference zoo.py:63-118): the qres34m backbone without its final patch_upsample; GaussianNLLOutputNet codes the
    3*H*W image samples per pixel (1.18 M symbols for 512x768) on top of the 12 latent strings => lossless."""
    ch = 96
    enc_nums, dec_nums, z_dims = [6, 6, 6, 4, 2], [1, 2, 3, 3, 3], [16, 14, 12, 10, 8]
    enc_k, dec_k = [7, 7, 5, 3, 1], [1, 3, 5, 7, 7]
    enc_w = [ch * 2, ch * 4, ch * 4, ch * 4, ch * 4]
    dec_w = [ch * 4, ch * 4, ch * 4, ch * 4, ch * 2]
    enc = [qres.StemParams(3, enc_w[0], 4)]
    for lvl in range(5):
        enc += [qres.MyCNXParams(enc_w[lvl], kernel_size=enc_k[lvl]) for _ in range(enc_nums[lvl])]
        if lvl < 4:
            enc.append(qres.MyCNXDownParams(enc_w[lvl], enc_w[lvl + 1]))
    dec = []
    for lvl in range(5):
        dec += [qres.QLBParams(dec_w[lvl], z_dims[lvl], kernel_size=dec_k[lvl]) for _ in range(dec_nums[lvl])]
        if lvl < 4:
            dec.append(UpParams(dec_w[lvl], dec_w[lvl + 1], 2))
    cfg = dict(enc_blocks=enc, dec_blocks=dec, out_net=qres.GaussianNLLOutParams(ch * 2, 3, rate=4),
               im_shift=-0.4546259594901961, im_scale=3.67572653978347, max_stride=64)
    model = qres.HierarchicalVAE(cfg)
    if pretrained is True:
        from torch.hub import load_state_dict_from_url
        url = 'https://huggingface.co/duanzh0/my-model-weights/resolve/main/qres34m/qres34m-lossless.pt'
        model.load_state_dict(load_state_dict_from_url(url)['model'])
    elif isinstance(pretrained, str):
        model.load_state_dict(torch.load(pretrained)['model'])
    else:
        assert pretrained is False, f'Invalid {pretrained=}'
    return model


@register_model
def qres17m(lmb=8, pretrained=False):
    """(reference zoo.py:121-166): 17 M parameters, ch = 72; strides 4/8/16/64 (the last down-sampling is 4x4/s4), decoder with a
    nearest x4 Upsample and two stride-2 transposed convs (k = 3, 5), 12 latent blocks z = 16 | 8,8 | 6 x4 | 4 x5; CelebA statistics."""
    ch = 72
    enc_nums, dec_nums, z_dims = [6, 6, 4, 2], [1, 2, 4, 5], [16, 8, 6, 4]
    enc = [qres.StemParams(3, ch * 2, 4)]
    enc += [qres.MyCNXParams(ch * 2, kernel_size=7) for _ in range(enc_nums[0])]
    enc.append(qres.MyCNXDownParams(ch * 2, ch * 4))
    enc += [qres.MyCNXParams(ch * 4, kernel_size=5) for _ in range(enc_nums[1])]
    enc.append(qres.MyCNXDownParams(ch * 4, ch * 4))
    enc += [qres.MyCNXParams(ch * 4, kernel_size=3) for _ in range(enc_nums[2])]
    enc.append(qres.MyCNXDownParams(ch * 4, ch * 4, down_rate=4))
    enc += [qres.MyCNXParams(ch * 4, kernel_size=1) for _ in range(enc_nums[3])]
    dec = [qres.QLBParams(ch * 4, z_dims[0], kernel_size=1) for _ in range(dec_nums[0])]
    dec.append(qres.NearestUpParams(4))
    dec += [qres.QLBParams(ch * 4, z_dims[1], kernel_size=3) for _ in range(dec_nums[1])]
    dec.append(qres.DeconvParams(ch * 4, ch * 4, kernel_size=3))
    dec += [qres.QLBParams(ch * 4, z_dims[2], kernel_size=5) for _ in range(dec_nums[2])]
    dec.append(qres.DeconvParams(ch * 4, ch * 2))
    dec += [qres.QLBParams(ch * 2, z_dims[3], kernel_size=7) for _ in range(dec_nums[3])]
    dec.append(UpParams(ch * 2, 3, 4))
    cfg = dict(enc_blocks=enc, dec_blocks=dec, im_shift=-0.4356, im_scale=3.397893306150187, max_stride=64)
    model = qres.HierarchicalVAE(cfg)
    model.mse_lmb = float(lmb)
    if (pretrained is True) and (lmb in {1, 2, 4, 8, 16, 32, 64, 1024}):
        from torch.hub import load_state_dict_from_url
        url = f'https://huggingface.co/duanzh0/my-model-weights/resolve/main/qres17m/qres17m-lmb{lmb}.pt'
        model.load_state_dict(load_state_dict_from_url(url)['model'])
    elif isinstance(pretrained, str):
        model.load_state_dict(torch.load(pretrained)['model'])
    else:
        assert pretrained is False, f'Invalid {pretrained=} and {lmb=}'
    return model
