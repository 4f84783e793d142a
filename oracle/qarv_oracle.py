"""oracle/qarv_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 CPU ops, NCHW, no timm/compressai/torchvision) of the reference's
QARV encode/decode hot path, driven by a state dict that uses the reference's own key names.  It is
the checker for the HIP path (tests/, __graft_entry__.smoke()) and the `cpu_baseline` leg of bench.py
("port": the reference's Python cannot travel to the GPU box).  The product package never imports it.

Pinned by: tests/golden/*.npz -- outputs of the reference's own classes imported in the build
container (tests/golden/make_golden.py); see tests/test_oracle_golden.py.  The entropy-coding part
(CompressAI semantics) is "parity unpinned" -- see compressai_semantics.py / rans_oracle.c headers.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math
import struct

import numpy as np
import torch
import torch.nn.functional as F

from . import compressai_semantics as cs


# ---------------------------------------------------------------- architecture (lvae/models/qarv/zoo.py:9-88)
def qarv_base_arch():
    ch = 128
    enc_dims = [192, ch * 3, ch * 4, ch * 4, ch * 4]
    enc = [('down', 3, enc_dims[0], 4)]
    enc += [('cnx', enc_dims[0], 7, 2)] * 7                      # zoo.py:39-40 (res_block default k=7)
    enc += [('down', enc_dims[0], enc_dims[1], 2)]
    enc += [('cnx', enc_dims[1], 7, 2)] * 6 + [('key', 'enc_s8'), ('cnx', enc_dims[1], 7, 2)]
    enc += [('down', enc_dims[1], enc_dims[2], 2)]
    enc += [('cnx', enc_dims[2], 5, 2)] * 6 + [('key', 'enc_s16'), ('cnx', enc_dims[2], 7, 2)]
    enc += [('down', enc_dims[2], enc_dims[3], 2)]
    enc += [('cnx', enc_dims[3], 3, 2)] * 4 + [('key', 'enc_s32'), ('cnx', enc_dims[3], 7, 2)]
    enc += [('down', enc_dims[3], enc_dims[4], 2)]
    enc += [('cnx', enc_dims[4], 1, 2)] * 4 + [('key', 'enc_s64')]
    dec_dims = [ch * 4, ch * 4, ch * 3, ch * 2, ch * 1]
    z = [32, 32, 96, 8]
    dec = [('vrlv', dec_dims[0], z[0], 'enc_s64', enc_dims[4], 1, 4), ('cnx', dec_dims[0], 1, 4),
           ('up', dec_dims[0], dec_dims[1], 2)]
    dec += [('cnx', dec_dims[1], 3, 3)] + [('vrlv', dec_dims[1], z[1], 'enc_s32', enc_dims[3], 3, 3)] * 2
    dec += [('cnx', dec_dims[1], 3, 3), ('up', dec_dims[1], dec_dims[2], 2)]
    dec += [('cnx', dec_dims[2], 5, 2)] + [('vrlv', dec_dims[2], z[2], 'enc_s16', enc_dims[2], 5, 2)] * 3
    dec += [('cnx', dec_dims[2], 5, 2), ('up', dec_dims[2], dec_dims[3], 2)]
    dec += [('cnx', dec_dims[3], 7, 1.75)] + [('vrlv', dec_dims[3], z[3], 'enc_s8', enc_dims[1], 7, 1.75)] * 3
    dec += [('stop',), ('cnx', dec_dims[3], 7, 1.75), ('up', dec_dims[3], dec_dims[4], 2)]
    dec += [('cnx', dec_dims[4], 7, 1.5)] * 8 + [('up', dec_dims[4], 3, 4)]
    return dict(enc=enc, dec=dec, im_shift=-0.4546259594901961, im_scale=3.67572653978347, max_stride=64,
                lmb_range=(16.0, 2048.0), lmb_embed_dim=(256, 256), sin_period=64, max_lmb=8192)


def cnx_shapes(prefix, dim, k, mlp_ratio, embed=256):
    hid = int(mlp_ratio * dim)
    return [(f'{prefix}.conv_dw.weight', (dim, 1, k, k)), (f'{prefix}.conv_dw.bias', (dim,)),
            (f'{prefix}.embedding_layer.1.weight', (2 * dim, embed)), (f'{prefix}.embedding_layer.1.bias', (2 * dim,)),
            (f'{prefix}.mlp.fc1.weight', (hid, dim)), (f'{prefix}.mlp.fc1.bias', (hid,)),
            (f'{prefix}.mlp.fc2.weight', (dim, hid)), (f'{prefix}.mlp.fc2.bias', (dim,)),
            (f'{prefix}.gamma', (1, dim, 1, 1))]


def qarv_param_shapes(arch):
    """(name, shape) of every parameter, reference key names."""
    out = []
    for i, b in enumerate(arch['enc']):
        p = f'encoder.enc_blocks.{i}'
        if b[0] == 'down':
            out += [(f'{p}.weight', (b[2], b[1], b[3], b[3])), (f'{p}.bias', (b[2],))]
        elif b[0] == 'cnx':
            out += cnx_shapes(p, b[1], b[2], b[3])
    for i, b in enumerate(arch['dec']):
        p = f'dec_blocks.{i}'
        if b[0] == 'cnx':
            out += cnx_shapes(p, b[1], b[2], b[3])
        elif b[0] == 'up':
            out += [(f'{p}.0.weight', (b[2] * b[3] ** 2, b[1], 1, 1)), (f'{p}.0.bias', (b[2] * b[3] ** 2,))]
        elif b[0] == 'vrlv':
            _, w, zd, _, ew, k, mlp = b
            out += cnx_shapes(f'{p}.resnet_front', w, k, mlp) + cnx_shapes(f'{p}.resnet_end', w, k, mlp)
            out += cnx_shapes(f'{p}.posterior0', ew, k, 2) + cnx_shapes(f'{p}.posterior1', w, k, 2)
            out += cnx_shapes(f'{p}.posterior2', w, k, 2)
            out += [(f'{p}.post_merge.weight', (w, w + ew, 1, 1)), (f'{p}.post_merge.bias', (w,)),
                    (f'{p}.posterior.weight', (zd, w, 3, 3)), (f'{p}.posterior.bias', (zd,)),
                    (f'{p}.z_proj.weight', (w, zd, 1, 1)), (f'{p}.z_proj.bias', (w,)),
                    (f'{p}.prior.weight', (2 * zd, w, 1, 1)), (f'{p}.prior.bias', (2 * zd,))]
    width = arch['dec'][0][1]
    out += [('bias', (1, width, 1, 1))]
    e0, e1 = arch['lmb_embed_dim']
    out += [('lmb_embedding.0.weight', (e1, e0)), ('lmb_embedding.0.bias', (e1,)),
            ('lmb_embedding.2.weight', (e1, e1)), ('lmb_embedding.2.bias', (e1,))]
    return out


# ---------------------------------------------------------------- entropy model (lvae/models/entropy_coding.py:52-82)
class DiscretizedGaussianOracle(cs.GaussianConditional):
    """Same construction as the reference subclass: EntropyModel ctor only (entropy_coding.py:60),
    64-entry log table 0.11..20 (:72-75), tail_mass 1e-9 (:67), LowerBound(table[0]) (:68),
    erf-form fp32 CDF = td.Normal(0,1).cdf (:70,81-82), scipy ppf quantile (:77-79)."""
    def __init__(self):
        cs.EntropyModel.__init__(self)
        scale_table = cs.log_spaced_table(0.11, 20.0, 64)        # entropy_coding.py:72-75, host-independent bits
        self.register_buffer('scale_table', scale_table, persistent=False)
        self.tail_mass = float(1e-9)
        self.lower_bound_scale = cs.LowerBound(scale_table[0])

    def _standardized_cumulative(self, inputs):
        # torch.distributions.Normal(0,1).cdf: 0.5 * (1 + erf((x - loc) * scale.reciprocal() / sqrt(2)))
        loc, scale = torch.tensor(0.0), torch.tensor(1.0)
        return 0.5 * (1 + torch.erf((inputs - loc) * scale.reciprocal() / math.sqrt(2)))


# ---------------------------------------------------------------- building blocks
def cnx_adaln(sd, p, x, emb):
    """ConvNeXtBlockAdaLN.forward (lvae/models/common.py:142-161)."""
    w = sd[f'{p}.conv_dw.weight']
    k = w.shape[-1]
    y = F.conv2d(x, w, sd[f'{p}.conv_dw.bias'], padding=(k - 1) // 2, groups=w.shape[0])   # :145
    y = y.permute(0, 2, 3, 1).contiguous()                                                  # :147
    y = F.layer_norm(y, (y.shape[-1],), eps=1e-6)                                           # :148 (no affine, :119)
    e = F.linear(F.gelu(emb), sd[f'{p}.embedding_layer.1.weight'], sd[f'{p}.embedding_layer.1.bias'])
    e = e.unflatten(1, (1, 1, e.shape[1]))                                                  # :123-127,150
    shift, scale = torch.chunk(e, 2, dim=-1)                                                # :151
    y = y * (1 + scale) + shift                                                             # :152
    y = F.linear(F.gelu(F.linear(y, sd[f'{p}.mlp.fc1.weight'], sd[f'{p}.mlp.fc1.bias'])),
                 sd[f'{p}.mlp.fc2.weight'], sd[f'{p}.mlp.fc2.bias'])                        # :154 (timm Mlp)
    y = y.permute(0, 3, 1, 2).contiguous()                                                  # :155
    y = y.mul(sd[f'{p}.gamma'])                                                             # :157-158
    return y + x                                                                            # :159-160


def conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[f'{p}.weight'], sd[f'{p}.bias'], stride=stride, padding=padding)


def lmb_embedding(sd, arch, lmb, n):
    """_get_lmb_embedding (qarv/model.py:281-287), _lmb_scaling (:275-279), sinusoidal_embedding (common.py:101-107)."""
    lmb_t = torch.full((n,), float(lmb))                                                    # :266-273
    scaled = torch.log(lmb_t) * arch['sin_period'] / math.log(arch['max_lmb'])
    dim = arch['lmb_embed_dim'][0]
    exponents = torch.linspace(0, 1, steps=dim // 2)
    freqs = torch.pow(arch['sin_period'], -1.0 * exponents)
    args = scaled.view(-1, 1) * freqs.view(1, dim // 2)
    e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    e = F.linear(e, sd['lmb_embedding.0.weight'], sd['lmb_embedding.0.bias'])
    e = F.gelu(e)
    return F.linear(e, sd['lmb_embedding.2.weight'], sd['lmb_embedding.2.bias'])            # :206-210


class QarvOracle:
    """VariableRateLossyVAE inference path (lvae/models/qarv/model.py:169-581)."""

    def __init__(self, state_dict, arch=None):
        self.arch = arch or qarv_base_arch()
        self.sd = {k: (torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v).float().cpu()
                   for k, v in state_dict.items() if 'discrete_gaussian' not in k}
        self.dg = DiscretizedGaussianOracle()
        self.default_lmb = self.arch['lmb_range'][1]                                        # :204
        self.max_stride = self.arch['max_stride']
        self.num_latents = sum(1 for b in self.arch['dec'] if b[0] == 'vrlv')

    def compress_mode(self, mode=True):                                                     # :509-514
        if mode:
            self.dg.update()        # all 9 blocks hold identical tables (same class, same scale table)

    # ---- pieces
    def encoder(self, x, emb):                                                              # common.py:89-98
        feats = {}
        for i, b in enumerate(self.arch['enc']):
            p = f'encoder.enc_blocks.{i}'
            if b[0] == 'key':
                feats[b[1]] = x
            elif b[0] == 'cnx':
                x = cnx_adaln(self.sd, p, x, emb)
            else:
                x = conv(self.sd, p, x, stride=b[3])                                        # common.py:29-30
        return feats

    def transform_prior(self, p, feature, emb):                                             # qarv/model.py:44-54
        feature = cnx_adaln(self.sd, f'{p}.resnet_front', feature, emb)
        pm, plogv = conv(self.sd, f'{p}.prior', feature).chunk(2, dim=1)
        plogv = F.softplus(plogv + 2.3) - 2.3
        return feature, pm, torch.exp(plogv)

    def transform_posterior(self, p, feature, enc_feature, emb):                            # :56-70
        e = cnx_adaln(self.sd, f'{p}.posterior0', enc_feature, emb)
        f = cnx_adaln(self.sd, f'{p}.posterior1', feature, emb)
        m = conv(self.sd, f'{p}.post_merge', torch.cat([f, e], dim=1))
        m = cnx_adaln(self.sd, f'{p}.posterior2', m, emb)
        return conv(self.sd, f'{p}.posterior', m, padding=1)

    def upsample(self, p, x, rate):                                                         # common.py:33-38
        return F.pixel_shuffle(conv(self.sd, f'{p}.0', x), rate)

    # ---- encode: returns per-block dicts (pm, pv, qm, indexes, symbols, z) and optionally strings
    @torch.no_grad()
    def encode_trace(self, im, lmb=None, code=True):
        lmb = lmb or self.default_lmb
        a = self.arch
        assert im.shape[2] % a['max_stride'] == 0 and im.shape[3] % a['max_stride'] == 0    # :219
        x = im.clone().add_(a['im_shift']).mul_(a['im_scale'])                              # :221
        nB = im.shape[0]
        emb = lmb_embedding(self.sd, a, lmb, nB)
        feats = self.encoder(x, emb)
        feature = self.sd['bias'].expand(nB, -1, x.shape[2] // 64, x.shape[3] // 64)        # :289-292,301
        blocks = []
        for i, b in enumerate(a['dec']):
            p = f'dec_blocks.{i}'
            if b[0] == 'vrlv':
                feature, pm, pv = self.transform_prior(p, feature, emb)                     # :85
                qm = self.transform_posterior(p, feature, feats[b[3]], emb)                 # :105
                indexes = self.dg.build_indexes(pv)                                         # :106
                symbols = self.dg.quantize(qm, 'symbols', pm)
                z = self.dg.quantize(qm, 'dequantize', means=pm)                            # :108
                rec = dict(pm=pm, pv=pv, qm=qm, indexes=indexes, symbols=symbols, z=z)
                if code:
                    rec['strings'] = self.dg.compress(qm, indexes, means=pm)                # :107
                blocks.append(rec)
                feature = feature + conv(self.sd, f'{p}.z_proj', z)                         # :72-75,117
                feature = cnx_adaln(self.sd, f'{p}.resnet_end', feature, emb)               # :118
            elif b[0] == 'cnx':
                feature = cnx_adaln(self.sd, p, feature, emb)
            elif b[0] == 'stop':
                break                                                                       # :310-312
            else:
                feature = self.upsample(p, feature, b[3])
        return dict(emb=emb, enc_features=feats, blocks=blocks)

    @torch.no_grad()
    def compress(self, im, lmb=None):                                                       # :516-529
        lmb = lmb or self.default_lmb
        tr = self.encode_trace(im, lmb, code=True)
        assert im.shape[0] == 1
        strings = [blk['strings'][0] for blk in tr['blocks']]
        body = pack_byte_strings(strings)
        nB, _, imH, imW = im.shape
        return struct.pack('f', lmb) + struct.pack('3H', nB, imH // 64, imW // 64) + body

    # ---- decode
    @torch.no_grad()
    def decode_from_latents(self, lmb, latents, bhw_repeat=None):
        """conditional_sample with given latents (qarv/model.py:365-395, branch :101-103): the decoder
        output for known z; identical to what decompress() reconstructs.
        A latent given as None is drawn from the prior at temperature t = 0, i.e. z = pm (:98-100 with t = 0):
        the deterministic form scripts/qarv/robust-decoding.py uses for progressive decoding."""
        if latents[0] is None:
            nB, nH, nW = bhw_repeat
        else:
            nB, _, nH, nW = latents[0].shape
        emb = lmb_embedding(self.sd, self.arch, lmb, nB)
        feature = self.sd['bias'].expand(nB, -1, nH, nW)
        li = 0
        for i, b in enumerate(self.arch['dec']):
            p = f'dec_blocks.{i}'
            if b[0] == 'vrlv':
                feature, pm, pv = self.transform_prior(p, feature, emb)
                feature = feature + conv(self.sd, f'{p}.z_proj', pm if latents[li] is None else latents[li])
                li += 1
                feature = cnx_adaln(self.sd, f'{p}.resnet_end', feature, emb)
            elif b[0] == 'cnx':
                feature = cnx_adaln(self.sd, p, feature, emb)
            elif b[0] == 'up':
                feature = self.upsample(p, feature, b[3])
        return feature.clone().clamp_(min=-1.0, max=1.0).mul_(0.5).add_(0.5)               # :224-232

    @torch.no_grad()
    def decompress(self, string):                                                           # :531-557
        lmb = struct.unpack('f', string[:4])[0]
        nB, nH, nW = struct.unpack('3H', string[4:10])
        strings = unpack_byte_string(string[10:])
        emb = lmb_embedding(self.sd, self.arch, lmb, nB)
        feature = self.sd['bias'].expand(nB, -1, nH, nW)
        si = 0
        for i, b in enumerate(self.arch['dec']):
            p = f'dec_blocks.{i}'
            if b[0] == 'vrlv':
                feature, pm, pv = self.transform_prior(p, feature, emb)
                indexes = self.dg.build_indexes(pv)                                         # :112
                z = self.dg.decompress([strings[si]], indexes, means=pm)                    # :113
                si += 1
                feature = feature + conv(self.sd, f'{p}.z_proj', z)
                feature = cnx_adaln(self.sd, f'{p}.resnet_end', feature, emb)
            elif b[0] == 'cnx':
                feature = cnx_adaln(self.sd, p, feature, emb)
            elif b[0] == 'up':
                feature = self.upsample(p, feature, b[3])
        assert si == len(strings)
        return feature.clone().clamp_(min=-1.0, max=1.0).mul_(0.5).add_(0.5)


# ---------------------------------------------------------------- bitstream container (lvae/utils/coding.py:26-70)
def pack_byte_strings(list_of_strings):
    lengths = [len(s) for s in list_of_strings]
    packed = b''.join(list_of_strings)
    packed = struct.pack(f'{len(lengths)}I', *lengths) + packed
    return struct.pack('B', len(lengths)) + packed


def unpack_byte_string(string):
    num, string = struct.unpack('B', string[:1])[0], string[1:]
    lengths, string = struct.unpack(f'{num}I', string[:num * 4]), string[num * 4:]
    assert sum(lengths) == len(string)
    edges = np.cumsum((0,) + lengths, dtype=np.int64)
    return [string[edges[i]:edges[i + 1]] for i in range(num)]


def pad_divisible_by_u8(img_hwc, div=64):
    """coding.pad_divisible_by (lvae/utils/coding.py:73-91) on a uint8 HxWxC array: replicate right/bottom."""
    h, w = img_hwc.shape[:2]
    ht, wt = div * math.ceil(h / div), div * math.ceil(w / div)
    if (ht, wt) == (h, w):
        return img_hwc
    return np.pad(img_hwc, ((0, ht - h), (0, wt - w), (0, 0)), mode='edge')
