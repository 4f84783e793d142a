"""oracle/qres_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch fp32 CPU ops, NCHW) of the reference's QRes-VAE (`qres34m`) encode/decode path, driven by a
state dict with the reference's key names.  Checker for the HIP path; never imported by the product.
Pinned by tests/golden/qres34m_*.npz (outputs of the reference's own classes, tests/golden/make_golden.py).
Entropy coding = stock CompressAI `GaussianConditional(None)` semantics (erfc-form CDF, scale_bound 0.11): parity
unpinned against real CompressAI, see compressai_semantics.py.

Cites are relative to /root/reference/lvae/models/qresvae/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import compressai_semantics as cs


def qres34m_arch():
    """zoo.py:9-48."""
    ch = 96
    enc_nums, dec_nums, z_dims = [6, 6, 6, 4, 2], [1, 2, 3, 3, 3], [16, 14, 12, 10, 8]
    enc = [('down', 3, ch * 2, 4)]
    enc += [('cnx', ch * 2, 7)] * enc_nums[0] + [('cnxdown', ch * 2, ch * 4, 7)]
    enc += [('cnx', ch * 4, 7)] * enc_nums[1] + [('cnxdown', ch * 4, ch * 4, 7)]
    enc += [('cnx', ch * 4, 5)] * enc_nums[2] + [('cnxdown', ch * 4, ch * 4, 7)]
    enc += [('cnx', ch * 4, 3)] * enc_nums[3] + [('cnxdown', ch * 4, ch * 4, 7)]
    enc += [('cnx', ch * 4, 1)] * enc_nums[4]
    dec = []
    widths = [ch * 4, ch * 4, ch * 4, ch * 4, ch * 2]
    ks = [1, 3, 5, 7, 7]
    for lvl in range(5):
        dec += [('qlb', widths[lvl], z_dims[lvl], ks[lvl])] * dec_nums[lvl]
        if lvl < 4:
            dec += [('up', widths[lvl], widths[lvl + 1], 2)]
    dec += [('up', ch * 2, 3, 4)]
    return dict(enc=enc, dec=dec, im_shift=-0.4546259594901961, im_scale=3.67572653978347, max_stride=64)


def qres17m_arch():
    """zoo.py:121-166."""
    ch = 72
    enc = [('down', 3, ch * 2, 4)]
    enc += [('cnx', ch * 2, 7)] * 6 + [('cnxdown', ch * 2, ch * 4, 7)]
    enc += [('cnx', ch * 4, 5)] * 6 + [('cnxdown', ch * 4, ch * 4, 7)]
    enc += [('cnx', ch * 4, 3)] * 4 + [('cnxdown', ch * 4, ch * 4, 7, 4)]
    enc += [('cnx', ch * 4, 1)] * 2
    dec = [('qlb', ch * 4, 16, 1)] + [('nearest', 4)]
    dec += [('qlb', ch * 4, 8, 3)] * 2 + [('deconv', ch * 4, ch * 4, 3)]
    dec += [('qlb', ch * 4, 6, 5)] * 4 + [('deconv', ch * 4, ch * 2, 5)]
    dec += [('qlb', ch * 2, 4, 7)] * 5 + [('up', ch * 2, 3, 4)]
    return dict(enc=enc, dec=dec, im_shift=-0.4356, im_scale=3.397893306150187, max_stride=64)


def qres34m_lossless_arch():
    """zoo.py:63-118: the qres34m backbone without its last patch_upsample, plus GaussianNLLOutputNet."""
    a = qres34m_arch()
    a['dec'] = a['dec'][:-1]
    a['out_net'] = dict(cin=192, rate=4, bin_size=1 / 127.5)
    return a


def _cnx_shapes(p, dim, k, mlp_ratio=2):
    hid = int(mlp_ratio * dim)
    return [(f'{p}.conv_dw.weight', (dim, 1, k, k)), (f'{p}.conv_dw.bias', (dim,)), (f'{p}.norm.weight', (dim,)),
            (f'{p}.norm.bias', (dim,)), (f'{p}.mlp.fc1.weight', (hid, dim)), (f'{p}.mlp.fc1.bias', (hid,)),
            (f'{p}.mlp.fc2.weight', (dim, hid)), (f'{p}.mlp.fc2.bias', (dim,)), (f'{p}.gamma', (dim,))]


def _vd_shapes(p, cin, hid, cout, k3):
    kk = 3 if k3 else 1
    return [(f'{p}.c1.weight', (hid, cin, 1, 1)), (f'{p}.c1.bias', (hid,)), (f'{p}.c2.weight', (hid, hid, kk, kk)),
            (f'{p}.c2.bias', (hid,)), (f'{p}.c3.weight', (hid, hid, kk, kk)), (f'{p}.c3.bias', (hid,)),
            (f'{p}.c4.weight', (cout, hid, 1, 1)), (f'{p}.c4.bias', (cout,))]


def qres_param_shapes(arch):
    out = []
    for i, b in enumerate(arch['enc']):
        p = f'encoder.enc_blocks.{i}'
        if b[0] == 'down':
            out += [(f'{p}.weight', (b[2], b[1], b[3], b[3])), (f'{p}.bias', (b[2],))]
        elif b[0] == 'cnx':
            out += _cnx_shapes(p, b[1], b[2])
        else:
            r = b[4] if len(b) > 4 else 2
            out += _cnx_shapes(p, b[1], b[3]) + [(f'{p}.downsapmle.weight', (b[2], b[1], r, r)), (f'{p}.downsapmle.bias', (b[2],))]
    for i, b in enumerate(arch['dec']):
        p = f'decoder.dec_blocks.{i}'
        if b[0] == 'up':
            out += [(f'{p}.0.weight', (b[2] * b[3] ** 2, b[1], 1, 1)), (f'{p}.0.bias', (b[2] * b[3] ** 2,))]
        elif b[0] == 'deconv':
            out += [(f'{p}.weight', (b[1], b[2], b[3], b[3])), (f'{p}.bias', (b[2],))]
        elif b[0] == 'nearest':
            pass
        else:
            _, w, z, k = b
            hid, k3 = int(w * 0.25), k >= 3
            out += _cnx_shapes(f'{p}.resnet_front', w, k) + _cnx_shapes(f'{p}.resnet_end', w, k)
            out += _vd_shapes(f'{p}.posterior', 2 * w, hid, z, k3) + _vd_shapes(f'{p}.prior', w, hid, 2 * z, k3)
            kk = 3 if k3 else 1
            out += [(f'{p}.z_proj.0.weight', (hid // 2, z, kk, kk)), (f'{p}.z_proj.0.bias', (hid // 2,)),
                    (f'{p}.z_proj.2.weight', (w, hid // 2, 1, 1)), (f'{p}.z_proj.2.bias', (w,))]
    out += [('decoder.bias', (1, arch['dec'][0][1], 1, 1))]
    if 'out_net' in arch:
        o = arch['out_net']
        for n in ('conv_mean', 'conv_scale'):
            out += [(f'out_net.{n}.0.weight', (3 * o['rate'] ** 2, o['cin'], 1, 1)), (f'out_net.{n}.0.bias', (3 * o['rate'] ** 2,))]
    return out


def my_cnx(sd, p, x):
    """MyConvNeXtBlock.forward (model.py:168-182): dwconv -> LN(affine, eps 1e-6) -> fc1 -> GELU -> fc2 -> *gamma -> +x."""
    w = sd[f'{p}.conv_dw.weight']
    k = w.shape[-1]
    y = F.conv2d(x, w, sd[f'{p}.conv_dw.bias'], padding=(k - 1) // 2, groups=w.shape[0])
    y = y.permute(0, 2, 3, 1).contiguous()
    y = F.layer_norm(y, (y.shape[-1],), sd[f'{p}.norm.weight'], sd[f'{p}.norm.bias'], eps=1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[f'{p}.mlp.fc1.weight'], sd[f'{p}.mlp.fc1.bias'])),
                 sd[f'{p}.mlp.fc2.weight'], sd[f'{p}.mlp.fc2.bias'])
    y = y.permute(0, 3, 1, 2).contiguous()
    y = y.mul(sd[f'{p}.gamma'].reshape(1, -1, 1, 1))
    return y + x


def vdblock(sd, p, x):
    """VDBlock.forward, residual=False (model.py:143-149)."""
    def c(name, t):
        w = sd[f'{p}.{name}.weight']
        return F.conv2d(t, w, sd[f'{p}.{name}.bias'], padding=(w.shape[-1] - 1) // 2)
    h = c('c1', F.gelu(x))
    h = c('c2', F.gelu(h))
    h = c('c3', F.gelu(h))
    return c('c4', F.gelu(h))


def conv(sd, p, x, stride=1):
    w = sd[f'{p}.weight']
    return F.conv2d(x, w, sd[f'{p}.bias'], stride=stride, padding=(w.shape[-1] - 1) // 2 if stride == 1 else 0)


class QresOracle:
    """HierarchicalVAE inference path (model.py:457-725)."""

    def __init__(self, state_dict, arch=None):
        self.arch = arch or qres34m_arch()
        self.sd = {k: (torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v).float().cpu()
                   for k, v in state_dict.items() if 'discrete_gaussian' not in k}
        self.dg = cs.GaussianConditional(None)                                            # model.py:240
        self.max_stride = 64

    def compress_mode(self, mode=True):
        if mode:                                                                           # model.py:317-325
            scale_table = cs.log_spaced_table(0.1, 20, 64)
            self.dg.update_scale_table(scale_table)
            self.dg.update()
            if 'out_net' in self.arch:                                                     # GaussianNLLOutputNet.update (:59-67)
                self.out_dg = cs.GaussianConditional(None, scale_bound=0.11)
                lower = self.out_dg.lower_bound_scale.bound.item()
                self.out_dg.update_scale_table(cs.log_spaced_table(lower, 20, 128))
                self.out_dg.update()

    def encoder(self, x):                                                                  # model.py:200-207
        feats = {}
        for i, b in enumerate(self.arch['enc']):
            p = f'encoder.enc_blocks.{i}'
            if b[0] == 'down':
                x = conv(self.sd, p, x, stride=b[3])
            elif b[0] == 'cnx':
                x = my_cnx(self.sd, p, x)
            else:
                x = conv(self.sd, f'{p}.downsapmle', my_cnx(self.sd, p, x), stride=(b[4] if len(b) > 4 else 2))      # model.py:184-192
            feats[int(x.shape[2])] = x
        return feats

    def dec_other(self, p, b, feature):
        if b[0] == 'up':                                                                   # common.py:33-38
            return F.pixel_shuffle(conv(self.sd, f'{p}.0', feature), b[3])
        if b[0] == 'deconv':                                                               # common.py:40-45
            k = b[3]
            return F.conv_transpose2d(feature, self.sd[f'{p}.weight'], self.sd[f'{p}.bias'], stride=2, padding=k // 2, output_padding=1)
        if b[0] == 'nearest':                                                              # nn.Upsample(scale_factor=4)
            return F.interpolate(feature, scale_factor=b[1], mode='nearest')
        raise ValueError(b)

    def transform_prior(self, p, feature):                                                 # model.py:245-255
        feature = my_cnx(self.sd, f'{p}.resnet_front', feature)
        pm, plogv = vdblock(self.sd, f'{p}.prior', feature).chunk(2, dim=1)
        plogv = F.softplus(plogv + 2.3) - 2.3
        return feature, pm, plogv

    def z_proj(self, p, z):                                                                # model.py:235-239
        return conv(self.sd, f'{p}.z_proj.2', F.gelu(conv(self.sd, f'{p}.z_proj.0', z)))

    @torch.no_grad()
    def encode_trace(self, im, code=True):
        a = self.arch
        x = (im + a['im_shift']) * a['im_scale']                                           # model.py:484-494
        feats = self.encoder(x)
        min_res = min(feats.keys())
        feature = self.sd['decoder.bias'].expand(feats[min_res].shape)                     # model.py:426-427
        blocks = []
        for i, b in enumerate(a['dec']):
            p = f'decoder.dec_blocks.{i}'
            if b[0] == 'qlb':                                                              # model.py:327-344
                feature, pm, plogv = self.transform_prior(p, feature)
                qm = vdblock(self.sd, f'{p}.posterior', torch.cat([feature, feats[int(feature.shape[2])]], dim=1))
                pv = torch.exp(plogv)
                indexes = self.dg.build_indexes(pv)
                rec = dict(pm=pm, pv=pv, qm=qm, indexes=indexes, symbols=self.dg.quantize(qm, 'symbols', pm))
                if code:
                    rec['strings'] = self.dg.compress(qm, indexes, means=pm)
                zhat = self.dg.quantize(qm, mode='dequantize', means=pm)
                rec['z'] = zhat
                blocks.append(rec)
                feature = feature + self.z_proj(p, zhat)
                feature = my_cnx(self.sd, f'{p}.resnet_end', feature)
            else:
                feature = self.dec_other(p, b, feature)
        return dict(enc_features=feats, blocks=blocks, smallest=tuple(feats[min_res].shape), feature=feature)

    def _prepare_codec(self, feature, x=None):                                             # model.py:69-80
        o = self.arch['out_net']
        pm = F.pixel_shuffle(conv(self.sd, 'out_net.conv_mean.0', feature), o['rate'])
        pm = torch.round(pm * 127.5 + 127.5) / 127.5 - 1
        plogv = F.pixel_shuffle(conv(self.sd, 'out_net.conv_scale.0', feature), o['rate'])
        pm = pm / o['bin_size']
        plogv = plogv - math.log(o['bin_size'])
        if x is not None:
            x = x / o['bin_size']
        return pm, plogv, x

    @torch.no_grad()
    def compress(self, im, trace=None):                                                    # model.py:649-668
        tr = self.encode_trace(im, code=True)
        obj = [blk['strings'] for blk in tr['blocks']] + [tr['smallest']]
        if 'out_net' in self.arch:                                                         # :664-667, :82-86
            x_tgt = (im - 0.5) * 2.0                                                       # preprocess_target :506-515
            pm, plogv, x = self._prepare_codec(tr['feature'], x_tgt)
            indexes = self.out_dg.build_indexes(torch.exp(plogv))
            if trace is not None:
                trace.update(pm=pm, indexes=indexes, symbols=self.out_dg.quantize(x, 'symbols', pm))
            obj.append(self.out_dg.compress(x, indexes, means=pm))
        return obj

    @torch.no_grad()
    def decode_from_latents(self, latents):
        """cond_sample with every latent given (model.py:591-603 -> forward_with_latents :403-417, QLatentBlockX.forward_uncond
        :284-315 with `latent` set: z = latent): the decoder output for known z, what decompress() reconstructs."""
        nB, _, nH, nW = latents[0].shape
        feature = self.sd['decoder.bias'].expand(nB, -1, nH, nW)
        li = 0
        for i, b in enumerate(self.arch['dec']):
            p = f'decoder.dec_blocks.{i}'
            if b[0] == 'qlb':
                feature, pm, plogv = self.transform_prior(p, feature)
                feature = feature + self.z_proj(p, latents[li])
                li += 1
                feature = my_cnx(self.sd, f'{p}.resnet_end', feature)
            else:
                feature = self.dec_other(p, b, feature)
        assert li == len(latents) and 'out_net' not in self.arch
        return feature.clone().clamp_(min=-1.0, max=1.0).mul_(0.5).add_(0.5)               # model.py:496-504

    @torch.no_grad()
    def decompress(self, obj):                                                             # model.py:670-687, 440-454
        final = None
        if 'out_net' in self.arch:
            obj, final = obj[:-1], obj[-1]
        feature = self.sd['decoder.bias'].expand(obj[-1])
        si = 0
        for i, b in enumerate(self.arch['dec']):
            p = f'decoder.dec_blocks.{i}'
            if b[0] == 'qlb':                                                              # model.py:346-360
                feature, pm, plogv = self.transform_prior(p, feature)
                indexes = self.dg.build_indexes(torch.exp(plogv))
                zhat = self.dg.decompress(obj[si], indexes, means=pm)
                si += 1
                feature = feature + self.z_proj(p, zhat)
                feature = my_cnx(self.sd, f'{p}.resnet_end', feature)
            else:
                feature = self.dec_other(p, b, feature)
        assert si == len(obj) - 1
        if final is not None:                                                              # :88-94
            pm, plogv, _ = self._prepare_codec(feature)
            indexes = self.out_dg.build_indexes(torch.exp(plogv))
            feature = self.out_dg.decompress(final, indexes, means=pm) * self.arch['out_net']['bin_size']
        return feature.clone().clamp_(min=-1.0, max=1.0).mul_(0.5).add_(0.5)               # model.py:496-504
