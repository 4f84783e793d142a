"""QRes-VAE (`qres34m`, fixed-rate hierarchical VAE) inference codec on MI355X.

API surface of the reference's `HierarchicalVAE` (lvae/models/qresvae/model.py:457-725) for the encode/decode path:
`compress_mode`, `compress` (-> list), `decompress`, `compress_file` / `decompress_file` (pickle container, :690-725),
`max_stride`, nn.Module behaviour, reference state-dict key names (incl. the reference's `downsapmle` spelling).
As for QARV, the module tree only owns parameters; the network runs as native HIP launches recorded in plans:

  MyConvNeXtBlock (:168-182)      -> lvae_dwconv_ln_f32 (affine LN) + fc1/GELU GEMM + fc2/gamma/residual GEMM
  MyConvNeXtPatchDown (:184-192)  -> the same + A_PATCH2 GEMM
  VDBlock (:143-149)              -> 4 GEMMs: c1 with GELU-on-load of its input (optionally a fused torch.cat of two
                                     sources), c2/c3 as A_CONV3 (or plain for k<3), GELU fused in each epilogue
  z_proj (:235-239)               -> A_CONV3/plain GEMM + GELU, then 1x1 GEMM with the residual add into the feature
  prior/posterior heads, coder    -> lvae_prior_index_f32 / lvae_quantize_f32 / lvae_dequantize_f32 + host rANS
"""
import math
import pickle

import numpy as np
import torch
import torch.nn as nn

from ... import _native
from ...engine import Plan, ptr
from ...utils import coding
from ..base import CodecBase, PREC_CODE, on_model_device
from ..entropy_coding import DiscretizedGaussian, log_spaced_table, rans_decode_streams, rans_encode_streams
from ..qarv.model import UpParams, _conv


# ----------------------------------------------------------------------------------------------- parameter holders
class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class MyCNXParams(nn.Module):
    """timm ConvNeXtBlock as subclassed by the reference (:162-166): conv_dw, norm (affine LN), mlp, gamma (C,)."""
    kind = 'cnx'

    def __init__(self, dim, kernel_size=7, mlp_ratio=2):
        super().__init__()
        self.dim, self.kernel_size, self.hidden = dim, kernel_size, int(mlp_ratio * dim)
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, self.hidden)
        self.gamma = nn.Parameter(1e-6 * torch.ones(dim))


class DeconvParams(nn.ConvTranspose2d):
    """common.deconv (common.py:40-45): ConvTranspose2d(k, stride 2, padding k//2, output_padding 1): doubles the resolution."""
    kind = 'deconv'

    def __init__(self, cin, cout, kernel_size=5):
        super().__init__(cin, cout, kernel_size=kernel_size, stride=2, output_padding=1, padding=kernel_size // 2)
        self.cin, self.cout, self.k, self.rate = cin, cout, kernel_size, 2


class NearestUpParams(nn.Upsample):
    """torch.nn.Upsample(scale_factor=s), nearest (qres17m, zoo.py:143)."""
    kind = 'nearest'

    def __init__(self, scale_factor):
        super().__init__(scale_factor=scale_factor)
        self.rate = int(scale_factor)


class MyCNXDownParams(MyCNXParams):
    kind = 'cnxdown'

    def __init__(self, in_ch, out_ch, kernel_size=7, down_rate=2):
        super().__init__(in_ch, kernel_size)
        assert down_rate in (2, 4)
        self.downsapmle = _conv(in_ch, out_ch, down_rate, down_rate, 0)          # [sic] reference attribute name (:187)
        self.out_ch, self.down_rate = out_ch, down_rate


class StemParams(nn.Conv2d):
    kind = 'down'

    def __init__(self, cin, cout, rate):
        super().__init__(cin, cout, rate, rate, 0)
        self.bias.data.mul_(0.0)
        self.rate = rate


class VDParams(nn.Module):
    """VDBlock (:120-141): c1 1x1, c2/c3 3x3 (or 1x1), c4 1x1."""
    def __init__(self, cin, hid, cout, use_3x3, zero_last=False):
        super().__init__()
        k, p = (3, 1) if use_3x3 else (1, 0)
        self.c1, self.c2, self.c3, self.c4 = _conv(cin, hid, 1), _conv(hid, hid, k, 1, p), _conv(hid, hid, k, 1, p), _conv(hid, cout, 1)
        if zero_last:
            self.c4.weight.data.mul_(0.0)
        self.cin, self.hid, self.cout, self.k = cin, hid, cout, k


class QLBParams(nn.Module):
    """QLatentBlockX (:210-243)."""
    kind = 'qlb'

    def __init__(self, width, zdim, kernel_size=7):
        super().__init__()
        self.width, self.zdim, self.kernel_size = width, zdim, kernel_size
        hid = int(width * 0.25)
        use3 = kernel_size >= 3
        self.hid, self.k = hid, (3 if use3 else 1)
        self.resnet_front = MyCNXParams(width, kernel_size)
        self.resnet_end = MyCNXParams(width, kernel_size)
        self.posterior = VDParams(2 * width, hid, zdim, use3)
        self.prior = VDParams(width, hid, 2 * zdim, use3, zero_last=True)
        self.z_proj = nn.Sequential(_conv(zdim, hid // 2, self.k, 1, (self.k - 1) // 2), nn.GELU(), _conv(hid // 2, width, 1))
        self.discrete_gaussian = DiscretizedGaussian(scale_table=None, cdf_form='erfc', scale_bound=0.11, persistent_table=True)
        self.discrete_gaussian.register_buffer('scale_bound', torch.Tensor([0.11]))

    def residual_scaling(self, N):
        self.z_proj[2].weight.data.mul_(math.sqrt(1 / 3 * N))     # (:242-243), operator precedence as in the reference


class _Holder(nn.Module):
    pass


class GaussianNLLOutParams(nn.Module):
    """GaussianNLLOutputNet (:16-94) of qres34m_lossless: conv_mean / conv_scale = patch_upsample(cin, 3, rate 4), and the
    per-pixel entropy model (stock GaussianConditional, scale_bound 0.11, 128 log-spaced scales 0.11..20: update() :59-67)."""
    def __init__(self, cin, im_channels=3, rate=4, bin_size=1 / 127.5):
        super().__init__()
        from ..qarv.model import UpParams
        self.conv_mean, self.conv_scale = UpParams(cin, im_channels, rate), UpParams(cin, im_channels, rate)
        self.cin, self.rate, self.bin_size = cin, rate, bin_size
        self.discrete_gaussian = DiscretizedGaussian(scale_table=None, cdf_form='erfc', scale_bound=0.11, persistent_table=True)
        self.discrete_gaussian.register_buffer('scale_bound', torch.Tensor([0.11]))

    def update(self):
        table = log_spaced_table(0.11, 20, 128)
        self.discrete_gaussian.update_scale_table(table, force=True)


# ----------------------------------------------------------------------------------------------- packed weights
class _Packed:
    def __init__(self, model, dev):
        self.t = {}
        f32 = dict(device=dev, dtype=torch.float32)

        def put(name, t):
            self.t[name] = t.detach().to(**f32).contiguous()

        def cnx(p, m):
            C, k = m.dim, m.kernel_size
            put(p + '.dw_w', m.conv_dw.weight.reshape(C, k * k).t()); put(p + '.dw_b', m.conv_dw.bias)
            put(p + '.ln_w', m.norm.weight); put(p + '.ln_b', m.norm.bias)
            put(p + '.fc1_w', m.mlp.fc1.weight); put(p + '.fc1_b', m.mlp.fc1.bias)
            put(p + '.fc2_w', m.mlp.fc2.weight); put(p + '.fc2_b', m.mlp.fc2.bias)
            put(p + '.gamma', m.gamma.reshape(C))

        def convw(name, c, pad_in=0, pad_out=0):
            w, b = c.weight, c.bias
            if pad_in:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad_in))        # zero weights for padded input channels
            if pad_out:                                                         # zero rows: padded output channels are exactly 0
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, pad_out))
                b = torch.nn.functional.pad(b, (0, pad_out))
            put(name + '.w', w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))     # [Cout][(i,j,ci)]  (== [Cout][Cin] for 1x1)
            put(name + '.b', b)

        def vd(p, m):
            for n in ('c1', 'c2', 'c3', 'c4'):
                convw(f'{p}.{n}', getattr(m, n))

        for i, m in enumerate(model.encoder.enc_blocks):
            p = f'encoder.enc_blocks.{i}'
            if m.kind == 'down':
                put(p + '.w', m.weight.reshape(m.out_channels, -1).t()); put(p + '.b', m.bias)
            else:
                cnx(p, m)
                if m.kind == 'cnxdown' and m.down_rate == 2:
                    convw(p + '.downsapmle', m.downsapmle)
                elif m.kind == 'cnxdown':
                    # 4x4/s4 conv as two 2x2 patch gathers: an exact space-to-depth (identity weights, channel order (i, j, ci))
                    # and a 2x2/s2 conv over the 4C-channel map whose k index (I, J, i, j, ci) is input pixel (2I+i, 2J+j)
                    C, w = m.dim, m.downsapmle.weight                           # [Cout][C][4][4]
                    put(p + '.s2d.w', torch.eye(4 * C)); put(p + '.s2d.b', torch.zeros(4 * C))
                    w6 = w.reshape(w.shape[0], C, 2, 2, 2, 2)                   # [n][ci][I][i][J][j]
                    put(p + '.downsapmle.w', w6.permute(0, 2, 4, 3, 5, 1).reshape(w.shape[0], 16 * C))
                    put(p + '.downsapmle.b', m.downsapmle.bias)
        for i, m in enumerate(model.decoder.dec_blocks):
            p = f'decoder.dec_blocks.{i}'
            if m.kind == 'up':
                w, b = m[0].weight.reshape(m.cout * m.rate ** 2, m.cin), m[0].bias
                if m.cout > 3:
                    r2 = m.rate ** 2
                    w = w.reshape(m.cout, r2, m.cin).permute(1, 0, 2).reshape(r2 * m.cout, m.cin)
                    b = b.reshape(m.cout, r2).t().reshape(-1)
                put(p + '.w', w); put(p + '.b', b)
            elif m.kind == 'deconv':
                # stride-2 transposed conv as ONE 3x3-gather GEMM with a PixelShuffle store: output pixel (2a+py, 2b+px) reads the
                # input pixels (a+di, b+dj), di, dj in {-1, 0, 1}, through kernel tap (py + p - 2 di, px + p - 2 dj) when it exists
                k, pd, cin, cout = m.k, m.k // 2, m.cin, m.cout
                wt = m.weight                                                   # [Cin][Cout][k][k]
                wp = torch.zeros(2, 2, cout, 3, 3, cin, dtype=wt.dtype)
                for py in range(2):
                    for px in range(2):
                        for di in (-1, 0, 1):
                            for dj in (-1, 0, 1):
                                ky, kx = py + pd - 2 * di, px + pd - 2 * dj
                                if 0 <= ky < k and 0 <= kx < k:
                                    wp[py, px, :, di + 1, dj + 1, :] = wt[:, :, ky, kx].t()
                put(p + '.w', wp.reshape(4 * cout, 9 * cin)); put(p + '.b', m.bias.repeat(4))
            elif m.kind == 'nearest':
                C, r2 = m.channels, m.rate ** 2
                put(p + '.w', torch.eye(C).repeat(r2, 1)); put(p + '.b', torch.zeros(r2 * C))
            else:
                cnx(p + '.resnet_front', m.resnet_front); cnx(p + '.resnet_end', m.resnet_end)
                vd(p + '.posterior', m.posterior); vd(p + '.prior', m.prior)
                zp, hp = (m.zdim + 3) // 4 * 4, (m.hid // 2 + 3) // 4 * 4      # GEMM K must be a multiple of 4 (16-B operand loads)
                convw(p + '.z_proj.0', m.z_proj[0], pad_in=zp - m.zdim, pad_out=hp - m.hid // 2)
                convw(p + '.z_proj.2', m.z_proj[2], pad_in=hp - m.hid // 2)
        put('bias', model.decoder.bias.reshape(-1))
        on = model.out_net
        if isinstance(on, GaussianNLLOutParams):
            # one GEMM for conv_mean | conv_scale with the PixelShuffle folded into the row order: row (i*r + j)*6 + c is
            # mean channel c (c < 3) or scale channel c - 3 of sub-pixel (i, j); conv output channel c*r^2 + i*r + j (common.py:33-38)
            r2 = on.rate ** 2
            wm, ws = on.conv_mean[0].weight.reshape(3, r2, on.cin), on.conv_scale[0].weight.reshape(3, r2, on.cin)
            bm, bs = on.conv_mean[0].bias.reshape(3, r2), on.conv_scale[0].bias.reshape(3, r2)
            put('out_net.w', torch.cat([wm, ws], 0).permute(1, 0, 2).reshape(r2 * 6, on.cin))
            put('out_net.b', torch.cat([bm, bs], 0).t().reshape(-1))
            odg = on.discrete_gaussian
            self.out_scale_table = odg.scale_table.detach().to(**f32).contiguous()
            self.out_scale_bound = float(odg.lower_bound_scale.bound.item())
        dg = model._dg()
        self.scale_table = dg.scale_table.detach().to(**f32).contiguous()
        self.scale_bound = float(dg.lower_bound_scale.bound.item())
        self.device = dev

    def p(self, name):
        return self.t[name].data_ptr()

    def bf16_map(self, mode):
        """Built once per mode, under a lock: plans are recorded concurrently by the pipeline-group threads, and a second
        builder would free the first one's bf16 copies while its plan still points at them."""
        from ..base import bf16_weight_map, bf16x3_weight_map, f16x2_weight_map, f16x2k32_weight_map, _W16_LOCK
        with _W16_LOCK:
            if not hasattr(self, '_w16'):
                self._w16 = {}
            if mode not in self._w16:
                self._w16[mode] = {'bf16': bf16_weight_map, 'bf16x3': bf16x3_weight_map, 'f16x2': f16x2_weight_map,
                                   'f16x2k32': f16x2k32_weight_map}[mode](self.t)
        return self._w16[mode][0]


class _QresPlan(Plan):
    def __init__(self, model, pk, B, H, W, encode):
        super().__init__(pk.device)
        lib, self.pk, self.B = self.lib, pk, B
        if model._prec == 'fp8':
            raise NotImplementedError("the 'fp8' mode (bf16 activation storage + MX-fp8 GEMMs, BASELINE config 5) is built for qarv_base")
        self.prec = PREC_CODE[model._prec]
        self.prec_name = model._prec
        self.w16 = pk.bf16_map(model._prec) if self.prec else None
        self.w16_x3 = pk.bf16_map('bf16x3') if self.prec == 4 else None
        self.w16_k32 = pk.bf16_map('f16x2k32') if self.prec == 4 else None
        self.lat_shapes, self.idx_off, self.sym_off, self.cuts = [], [], [], []
        self.qcuts, self.prm_bufs, self.qm_bufs, self.zhat_bufs, self.zhat_ld = [], [], [], [], []   # test access (CodecBase._trace_blocks)
        nH, nW = H // 64, W // 64
        # latent I/O sizes: resolution doubles at every rate-2 upsample of the top-down path
        tot, s = 0, 1
        for m in model.decoder.dec_blocks:
            if m.kind == 'qlb':
                tot += m.zdim * nH * s * nW * s
            elif not (m.kind == 'up' and m.cout <= 3):
                s *= m.rate
        self.n_sym = tot * B
        self.sym_all, self.idx_all = self.new(self.n_sym, torch.int32), self.new(self.n_sym, torch.uint8)
        self.sym_host = torch.empty(self.n_sym, dtype=torch.int32).pin_memory()
        self.idx_host = torch.empty(self.n_sym, dtype=torch.uint8).pin_memory()
        self.sym_np, self.idx_np = self.sym_host.numpy(), self.idx_host.numpy()
        feats = {}
        if encode:
            self.im = self.new(B * 3 * H * W)
            h, w, x = H, W, None
            for i, m in enumerate(model.encoder.enc_blocks):
                p = f'encoder.enc_blocks.{i}'
                if m.kind == 'down':
                    h, w = h // 4, w // 4
                    x = self.new(B * h * w * m.out_channels)
                    self.add(lib.lvae_stem_f32, (self.im.data_ptr(), pk.p(p + '.w'), pk.p(p + '.b'), x.data_ptr(), B, H, W,
                                                 m.out_channels, model.im_shift, model.im_scale, self.status_ptr()), p + '.stem')
                elif m.kind == 'cnx':
                    self.cnx(p, m, x.data_ptr(), x.data_ptr(), h, w)
                else:           # CNX out of place (x is this level's encoder feature), then 2x2/s2 conv
                    t = self.buf('cnxdown_tmp', x.numel())
                    self.cnx(p, m, x.data_ptr(), t.data_ptr(), h, w)
                    feats[h] = x
                    h, w = h // 2, w // 2
                    if m.down_rate == 4:
                        s2d = self.buf('s2d', B * h * w * 4 * m.dim)
                        self.gemm(A0=t.data_ptr(), K0=m.dim, M=B * h * w, N=4 * m.dim, K=4 * m.dim, Wt=pk.p(p + '.s2d.w'),
                                  bias=pk.p(p + '.s2d.b'), out=s2d.data_ptr(), a_mode=_native.A_PATCH2, H=h, W=w, exact=True,
                                  label=p + '.space_to_depth')
                        h, w = h // 2, w // 2
                        nx = self.new(B * h * w * m.out_ch)
                        self.gemm(A0=s2d.data_ptr(), K0=4 * m.dim, M=B * h * w, N=m.out_ch, K=16 * m.dim, Wt=pk.p(p + '.downsapmle.w'),
                                  bias=pk.p(p + '.downsapmle.b'), out=nx.data_ptr(), a_mode=_native.A_PATCH2, H=h, W=w, label=p + '.down4')
                    else:
                        nx = self.new(B * h * w * m.out_ch)
                        self.gemm(A0=t.data_ptr(), K0=m.dim, M=B * h * w, N=m.out_ch, K=4 * m.dim, Wt=pk.p(p + '.downsapmle.w'),
                                  bias=pk.p(p + '.downsapmle.b'), out=nx.data_ptr(), a_mode=_native.A_PATCH2, H=h, W=w, label=p + '.down')
                    x = nx
            feats[h] = x
        # top-down path
        h, w = nH, nW
        width = model.decoder.dec_blocks[0].width
        f = self.new(B * h * w * width)
        self.add(lib.lvae_bias_expand_f32, (pk.p('bias'), f.data_ptr(), B * h * w, width), 'bias')
        self.out = None
        for i, m in enumerate(model.decoder.dec_blocks):
            p = f'decoder.dec_blocks.{i}'
            if m.kind == 'deconv':
                nf = self.new(B * h * w * 4 * m.cout)
                self.gemm(A0=f.data_ptr(), K0=m.cin, M=B * h * w, N=4 * m.cout, K=9 * m.cin, Wt=pk.p(p + '.w'), bias=pk.p(p + '.b'),
                          out=nf.data_ptr(), a_mode=_native.A_CONV3, store=_native.ST_SHUFFLE, r=2, H=h, W=w, label=p + '.deconv')
                f, h, w = nf, h * 2, w * 2
                continue
            if m.kind == 'nearest':
                C = m.channels
                nf = self.new(B * h * w * m.rate ** 2 * C)
                self.gemm(A0=f.data_ptr(), K0=C, M=B * h * w, N=m.rate ** 2 * C, Wt=pk.p(p + '.w'), bias=pk.p(p + '.b'), out=nf.data_ptr(),
                          store=_native.ST_SHUFFLE, r=m.rate, H=h, W=w, exact=True, label=p + '.nearest')
                f, h, w = nf, h * m.rate, w * m.rate
                continue
            if m.kind == 'up':
                nf = self.new(B * h * w * m.rate ** 2 * m.cout)
                final = m.cout <= 3
                self.gemm(A0=f.data_ptr(), K0=m.cin, M=B * h * w, N=m.cout * m.rate ** 2, Wt=pk.p(p + '.w'), bias=pk.p(p + '.b'),
                          out=nf.data_ptr(), store=_native.ST_IMAGE if final else _native.ST_SHUFFLE, r=m.rate, H=h, W=w, label=p + '.up')
                f, h, w = nf, h * m.rate, w * m.rate
                if final:
                    self.out = nf.view(B, m.cout, h, w)
                continue
            M, z, hid = B * h * w, m.zdim, m.hid
            zp = (z + 3) // 4 * 4
            self.cnx(p + '.resnet_front', m.resnet_front, f.data_ptr(), f.data_ptr(), h, w)
            prm = self.buf('prm', M * 2 * z)
            self.vdblock(p + '.prior', m.prior, f.data_ptr(), None, prm.data_ptr(), h, w)
            pm = self.new(M * z)
            ioff = sum(a * b for a, b in self.lat_shapes) * B
            self.lat_shapes.append((z, h * w)); self.idx_off.append(ioff); self.sym_off.append(ioff)
            self.add(lib.lvae_prior_index_f32, (prm.data_ptr(), pm.data_ptr(), ptr(self.idx_all, ioff), pk.scale_table.data_ptr(),
                                                pk.scale_table.numel(), pk.scale_bound, B, h * w, z, self.status_ptr()), p + '.prior_index')
            zhat = self.buf('zhat', M * zp)
            self.prm_bufs.append(prm); self.zhat_bufs.append(zhat); self.zhat_ld.append(zp)
            if encode:
                qm = self.buf('qm', M * z)
                self.vdblock(p + '.posterior', m.posterior, f.data_ptr(), feats[h].data_ptr(), qm.data_ptr(), h, w)
                self.add(lib.lvae_quantize_f32, (qm.data_ptr(), pm.data_ptr(), ptr(self.sym_all, ioff), zhat.data_ptr(), B, h * w, z, zp, self.status_ptr()),
                         p + '.quantize')
                self.qm_bufs.append(qm)
                self.qcuts.append(len(self.ops))
            else:
                self.cuts.append(len(self.ops))
                self.add(lib.lvae_dequantize_f32, (ptr(self.sym_all, ioff), pm.data_ptr(), zhat.data_ptr(), B, h * w, z, zp), p + '.dequantize')
            hp = (hid // 2 + 3) // 4 * 4
            v = self.buf('zproj_h', M * hp)
            conv3 = m.k == 3
            self.gemm(A0=zhat.data_ptr(), K0=zp, M=M, N=hp, K=(9 * zp if conv3 else zp), Wt=pk.p(p + '.z_proj.0.w'),
                      bias=pk.p(p + '.z_proj.0.b'), out=v.data_ptr(), a_mode=_native.A_CONV3 if conv3 else _native.A_PLAIN, H=h, W=w,
                      epi=_native.EPI_BIAS_GELU, label=p + '.z_proj.0')
            self.gemm(A0=v.data_ptr(), K0=hp, M=M, N=m.width, Wt=pk.p(p + '.z_proj.2.w'), bias=pk.p(p + '.z_proj.2.b'),
                      res=f.data_ptr(), ldres=m.width, out=f.data_ptr(), epi=_native.EPI_RES, label=p + '.z_proj.2')
            self.cnx(p + '.resnet_end', m.resnet_end, f.data_ptr(), f.data_ptr(), h, w)
        self.lossless = isinstance(model.out_net, GaussianNLLOutParams)
        if self.lossless:
            # GaussianNLLOutputNet.compress / decompress (:69-94): per-pixel coding of the 3*H*W image samples
            on = model.out_net
            Ho, Wo = h * on.rate, w * on.rate
            assert (Ho, Wo) == (H, W)
            raw = self.new(B * Ho * Wo * 6)
            self.px_raw = raw                               # test access: conv_mean | conv_scale after PixelShuffle, NHWC [B*H*W][6]
            self.gemm(A0=f.data_ptr(), K0=on.cin, M=B * h * w, N=6 * on.rate ** 2, Wt=pk.p('out_net.w'), bias=pk.p('out_net.b'),
                      out=raw.data_ptr(), store=_native.ST_SHUFFLE, r=on.rate, H=h, W=w, label='out_net.conv')
            npx = B * 3 * H * W
            self.px_pm = self.new(npx)
            self.px_sym, self.px_idx = self.new(npx, torch.int32), self.new(npx, torch.uint8)
            self.px_sym_host = torch.empty(npx, dtype=torch.int32).pin_memory()
            self.px_idx_host = torch.empty(npx, dtype=torch.uint8).pin_memory()
            self.px_sym_np, self.px_idx_np = self.px_sym_host.numpy(), self.px_idx_host.numpy()
            self.add(lib.lvae_lossless_params_f32, (raw.data_ptr(), self.im.data_ptr() if encode else None, self.px_pm.data_ptr(),
                                                    self.px_idx.data_ptr(), self.px_sym.data_ptr() if encode else None,
                                                    pk.out_scale_table.data_ptr(), pk.out_scale_table.numel(), pk.out_scale_bound,
                                                    B, H, W, self.status_ptr()), 'out_net.params')
            if not encode:
                self.cuts.append(len(self.ops))
                out = self.new(npx)
                self.add(lib.lvae_lossless_output_f32, (self.px_sym.data_ptr(), self.px_pm.data_ptr(), out.data_ptr(), npx, self.status_ptr()), 'out_net.output')
                self.out = out.view(B, 3, H, W)
        if not encode:
            assert self.out is not None

    def cnx(self, p, m, x, out, H, W):
        pk, lib = self.pk, self.lib
        C, k, hid = m.dim, m.kernel_size, m.hidden
        M = self.B * H * W
        if self.mlp_fused_ok(C, hid, k, M=M, rows_per_image=H * W):
            # C = 192 / hidden = 384 (the stride-4 blocks of qres34m, encoder and decoder): fc1 -> GELU -> fc2 as one launch (csrc/mlp_h2c.hip),
            # the bits of the two launches below
            y = self.buf('y', M * C)
            self.add(lib.lvae_dwconv_ln_h2, (x, pk.p(p + '.dw_w'), pk.p(p + '.dw_b'), pk.p(p + '.ln_w'), pk.p(p + '.ln_b'), None, None,
                                             y.data_ptr(), self.B, H, W, C, k), p + '.dwln')
            self.mlp_fused(y=y.data_ptr(), M=M, C=C, hid=hid, w1=pk.p(p + '.fc1_w'), b1=pk.p(p + '.fc1_b'), w2=pk.p(p + '.fc2_w'),
                           b2=pk.p(p + '.fc2_b'), gamma=pk.p(p + '.gamma'), res=x, out=out, label=p + '.mlp')
            return
        y, hbuf = self.buf('y', M * C), self.buf('hid', M * hid)
        pre1, pre2, S1, S2 = self.mlp_pipeline(C, hid, k, H * W)        # f16x2 plans: pre-split y / hidden map (see the qarv plan's cnx)
        self.add(lib.lvae_dwconv_ln_h2 if pre1 else lib.lvae_dwconv_ln_f32,
                 (x, pk.p(p + '.dw_w'), pk.p(p + '.dw_b'), pk.p(p + '.ln_w'), pk.p(p + '.ln_b'), None, None,
                  y.data_ptr(), self.B, H, W, C, k), p + '.dwln')
        self.gemm(A0=y.data_ptr(), K0=C, M=M, N=hid, Wt=pk.p(p + '.fc1_w'), bias=pk.p(p + '.fc1_b'), out=hbuf.data_ptr(),
                  epi=_native.EPI_BIAS_GELU, a_h2=pre1, out_h2=pre2, ksplit=S1, label=p + '.fc1')
        self.gemm(A0=hbuf.data_ptr(), K0=hid, M=M, N=C, Wt=pk.p(p + '.fc2_w'), bias=pk.p(p + '.fc2_b'), gamma=pk.p(p + '.gamma'),
                  res=x, ldres=C, out=out, epi=_native.EPI_GAMMA_RES, a_h2=pre2, ksplit=S2, label=p + '.fc2')

    def vdblock(self, p, m, a0, a1, out, H, W):
        """c4(g(c3(g(c2(g(c1(g(x)))))))) with x = a0 or cat[a0, a1] (each of width cin or cin/2)."""
        pk = self.pk
        M, hid = self.B * H * W, m.hid
        t1, t2 = self.buf('vd1', M * hid), self.buf('vd2', M * hid)
        k0 = m.cin if a1 is None else m.cin // 2
        self.gemm(A0=a0, K0=k0, A1=a1, K1=(0 if a1 is None else k0), lda1=(0 if a1 is None else k0), M=M, N=hid,
                  Wt=pk.p(p + '.c1.w'), bias=pk.p(p + '.c1.b'), out=t1.data_ptr(), a_gelu=1, epi=_native.EPI_BIAS_GELU, label=p + '.c1')
        mode = _native.A_CONV3 if m.k == 3 else _native.A_PLAIN
        kk = 9 * hid if m.k == 3 else hid
        self.gemm(A0=t1.data_ptr(), K0=hid, M=M, N=hid, K=kk, Wt=pk.p(p + '.c2.w'), bias=pk.p(p + '.c2.b'), out=t2.data_ptr(),
                  a_mode=mode, H=H, W=W, epi=_native.EPI_BIAS_GELU, label=p + '.c2')
        self.gemm(A0=t2.data_ptr(), K0=hid, M=M, N=hid, K=kk, Wt=pk.p(p + '.c3.w'), bias=pk.p(p + '.c3.b'), out=t1.data_ptr(),
                  a_mode=mode, H=H, W=W, epi=_native.EPI_BIAS_GELU, label=p + '.c3')
        self.gemm(A0=t1.data_ptr(), K0=hid, M=M, N=m.cout, Wt=pk.p(p + '.c4.w'), bias=pk.p(p + '.c4.b'), out=out, label=p + '.c4')


# ----------------------------------------------------------------------------------------------- the model
class HierarchicalVAE(CodecBase):
    log2_e = math.log2(math.e)

    def __init__(self, config: dict):
        super().__init__()
        self.encoder = _Holder()
        self.encoder.enc_blocks = nn.ModuleList(config.pop('enc_blocks'))
        self.decoder = _Holder()
        self.decoder.dec_blocks = nn.ModuleList(config.pop('dec_blocks'))
        width = self.decoder.dec_blocks[0].width
        cur = width
        for b in self.decoder.dec_blocks:                     # nn.Upsample has no channel count of its own
            if b.kind == 'nearest':
                b.channels = cur
            elif b.kind in ('up', 'deconv'):
                cur = b.cout
        self.decoder.bias = nn.Parameter(torch.zeros(1, width, 1, 1))
        n_res = len([b for b in self.decoder.dec_blocks if hasattr(b, 'residual_scaling')])
        for b in self.decoder.dec_blocks:                     # TopDownDecoder._init_weights (:373-377)
            if hasattr(b, 'residual_scaling'):
                b.residual_scaling(n_res)
        self.out_net = config.pop('out_net', nn.Identity())
        self.im_shift, self.im_scale = float(config['im_shift']), float(config['im_scale'])
        self.max_stride = config['max_stride']
        self.register_buffer('_dummy', torch.zeros(1), persistent=False)
        self.compressing = False
        self._packed, self._plans = None, {}
        self._init_codec_base()

    def _dg(self):
        for b in self.decoder.dec_blocks:
            if b.kind == 'qlb':
                return b.discrete_gaussian
        raise RuntimeError('no latent block')

    def _apply(self, fn, *a, **k):
        self._packed, self._plans = None, {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **k):
        """Reference key names; entropy-model buffers (`*.discrete_gaussian.*`) absent from / extra in a checkpoint are
        tolerated (their set differs between CompressAI versions; the tables are rebuilt by compress_mode())."""
        self._packed, self._plans = None, {}
        own = self.state_dict()
        sd = {kk: v for kk, v in state_dict.items() if not ('.discrete_gaussian.' in kk and (kk not in own or own[kk].shape != v.shape))}
        for kk, v in own.items():
            if '.discrete_gaussian.' in kk and kk not in sd:
                sd[kk] = v
        return super().load_state_dict(sd, strict=strict, **k)

    def _prepare(self):
        dev = self._dummy.device
        if self._packed is None or self._packed.device != dev:
            if dev.type != 'cuda':
                raise RuntimeError('lvae (MI355X build): compress/decompress run on the GPU only; move the model with '
                                   '.to("cuda") -- there is deliberately no CPU fallback')
            _native.lib()
            with torch.no_grad():
                self._packed = _Packed(self, dev)
            self._plans = {}
        return self._packed

    def _plan(self, kind, B, H, W, group=0):
        key = (kind, B, H, W, group, self._prec)
        pl = self._plans.get(key)
        if pl is None:
            pl = _QresPlan(self, self._prepare(), B, H, W, encode=(kind == 'enc'))
            self._plans[key] = pl
        return pl

    def compress_mode(self, mode=True):
        """(:640-647) -> QLatentBlockX.update (:317-325): 64 log-spaced scales 0.1..20, stock erfc-form tables."""
        if mode:
            table = log_spaced_table(0.1, 20, 64)
            first = None
            for b in self.decoder.dec_blocks:
                if b.kind != 'qlb':
                    continue
                dg = b.discrete_gaussian
                if first is None:
                    dg.update_scale_table(table, force=True)
                    first = dg
                else:
                    dg.scale_table = first.scale_table
                    dg._quantized_cdf, dg._offset, dg._cdf_length, dg._host = first._quantized_cdf, first._offset, first._cdf_length, None
            if isinstance(self.out_net, GaussianNLLOutParams):            # (:645-646)
                self.out_net.update()
            self._log_precision()
            # the packed device copy holds the scale table: one built before this call (encode_trace(), or a compress() that
            # stopped at 'Uninitialized CDFs') would keep the empty pre-update table
            self._packed, self._plans = None, {}
        self.compressing = mode

    @torch.no_grad()
    @on_model_device
    def compress_batch(self, im):
        """(B,3,H,W) -> list of B compressed objects, each `[ [bytes] x 12, (1, C, H/64, W/64) ]` as `compress()` returns."""
        assert im.dim() == 4 and im.shape[1] == 3
        B, _, H, W = im.shape
        assert H % self.max_stride == 0 and W % self.max_stride == 0, f'{im.shape=}'
        self._prepare()
        tables = self._dg().host_tables()
        groups = self._groups(B, 'enc')
        nthreads = self._coder_threads_per_group(len(groups))
        width = self.decoder.dec_blocks[0].width

        def encode_group(g, start, n, stream):
            pl = self._plan('enc', n, H, W, g)
            pl.im.view(n, 3, H, W).copy_(im[start:start + n])
            pl.run(stream=stream.cuda_stream)
            pl.sym_host.copy_(pl.sym_all, non_blocking=True)
            pl.idx_host.copy_(pl.idx_all, non_blocking=True)
            if pl.lossless:
                pl.px_sym_host.copy_(pl.px_sym, non_blocking=True)
                pl.px_idx_host.copy_(pl.px_idx, non_blocking=True)
            pl.fetch_status()
            stream.synchronize()
            pl.raise_if_flagged(where='while encoding')      # out-of-range input / NaN or inf in a prior parameter or posterior mean
            sv, iv = [], []
            for b in range(n):
                for li, (z, hw) in enumerate(pl.lat_shapes):
                    o = pl.sym_off[li] + b * z * hw
                    sv.append(pl.sym_np[o:o + z * hw]); iv.append(pl.idx_np[o:o + z * hw])
            strings = rans_encode_streams(tables, sv, iv, nthreads)
            nl = len(pl.lat_shapes)
            objs = [[[s] for s in strings[b * nl:(b + 1) * nl]] + [(1, width, H // 64, W // 64)] for b in range(n)]
            if pl.lossless:                                  # final string of the output net (:664-667)
                px = 3 * H * W
                fin = rans_encode_streams(self.out_net.discrete_gaussian.host_tables(),
                                          [pl.px_sym_np[b * px:(b + 1) * px] for b in range(n)],
                                          [pl.px_idx_np[b * px:(b + 1) * px] for b in range(n)], nthreads)
                for b in range(n):
                    objs[b].append([fin[b]])
            return objs

        out = []
        for part in self._run_groups(encode_group, groups):
            out += part
        return out

    @torch.no_grad()
    def compress(self, im):
        """(:649-668)."""
        assert im.shape[0] == 1, 'use compress_batch for more than one image'
        return self.compress_batch(im)[0]

    @torch.no_grad()
    @on_model_device
    def decompress_batch(self, objs):
        B = len(objs)
        lossless = isinstance(self.out_net, GaussianNLLOutParams)
        si = -2 if lossless else -1                          # position of the feature-shape tuple (:660-667)
        shape = tuple(objs[0][si])
        assert all(tuple(o[si]) == shape for o in objs) and shape[0] == 1
        nH, nW = shape[2], shape[3]
        H, W = nH * 64, nW * 64
        self._prepare()
        tables = self._dg().host_tables()
        groups = self._groups(B, 'dec')
        nthreads = self._coder_threads_per_group(len(groups))
        out = torch.empty(B, 3, H, W, device=self._dummy.device)

        def decode_group(g, start, n, stream):
            pl = self._plan('dec', n, H, W, g)
            assert all(len(objs[start + b]) - 1 == len(pl.cuts) for b in range(n)), 'wrong number of latent strings'
            if self.native_group_loops and not pl.lossless:
                # the loop below as ONE foreign call (csrc/plan_runtime.cpp::lvae_decode_blocks; lossless plans keep the loop: their
                # last stream uses the output net's tables)
                self._decode_group_native(pl, pl.cuts, pl.idx_off, n, [[objs[start + b][li][0] for li in range(len(pl.cuts))] for b in range(n)],
                                          tables, nthreads, stream)
                out[start:start + n].copy_(pl.out, non_blocking=True)
                return
            lo = 0
            for li, cut in enumerate(pl.cuts):
                pl.run(lo, cut, stream=stream.cuda_stream)
                lo = cut
                if pl.lossless and li == len(pl.cuts) - 1:   # the per-pixel stream of the output net (:680-682)
                    px = 3 * H * W
                    pl.px_idx_host.copy_(pl.px_idx, non_blocking=True)
                    stream.synchronize()
                    rans_decode_streams(self.out_net.discrete_gaussian.host_tables(), [objs[start + b][-1][0] for b in range(n)],
                                        [pl.px_idx_np[b * px:(b + 1) * px] for b in range(n)],
                                        [pl.px_sym_np[b * px:(b + 1) * px] for b in range(n)], nthreads)
                    pl.px_sym.copy_(pl.px_sym_host, non_blocking=True)
                    continue
                z, hw = pl.lat_shapes[li]
                o, cnt = pl.idx_off[li], n * z * hw
                pl.idx_host[o:o + cnt].copy_(pl.idx_all[o:o + cnt], non_blocking=True)
                stream.synchronize()
                iv = [pl.idx_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                sv = [pl.sym_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                rans_decode_streams(tables, [objs[start + b][li][0] for b in range(n)], iv, sv, nthreads)
                pl.sym_all[o:o + cnt].copy_(pl.sym_host[o:o + cnt], non_blocking=True)
            pl.run(lo, None, stream=stream.cuda_stream)
            pl.fetch_status()                               # read by _check_decoded() after the groups have finished
            out[start:start + n].copy_(pl.out, non_blocking=True)

        self._run_groups(decode_group, groups)
        self._check_decoded(groups, lambda g, n: self._plan('dec', n, H, W, g))
        return out

    @torch.no_grad()
    def decompress(self, compressed_object):
        """(:670-687)."""
        return self.decompress_batch([compressed_object])

    @torch.no_grad()
    def compress_file(self, img_path, output_path):
        """(:689-707): pickle of [strings..., feature shape, (h, w)]."""
        from PIL import Image
        img = Image.open(img_path)
        img_padded = coding.pad_divisible_by(img, div=self.max_stride)
        im = coding.pil_to_tensor01(img_padded).unsqueeze_(0).to(device=self._dummy.device)
        obj = self.compress(im)
        obj.append((img.height, img.width))
        with open(output_path, 'wb') as f:
            pickle.dump(obj, file=f)

    @torch.no_grad()
    def decompress_file(self, bits_path):
        """(:709-725)."""
        with open(bits_path, 'rb') as f:
            obj = pickle.load(file=f)
        img_h, img_w = obj.pop()
        return self.decompress(obj)[:, :, :img_h, :img_w]

    @torch.no_grad()
    def compress_files(self, img_paths, output_paths, images=None):
        """Batched compress_file (same padded size): one compress_batch call; files identical to compress_file's."""
        from PIL import Image
        imgs = images if images is not None else [Image.open(p) for p in img_paths]      # `images`: already decoded PIL images
        ims = [coding.pil_to_tensor01(coding.pad_divisible_by(img, div=self.max_stride)) for img in imgs]
        assert all(t.shape == ims[0].shape for t in ims), 'compress_files: padded sizes differ'
        objs = self.compress_batch(torch.stack(ims).to(device=self._dummy.device))
        for img, obj, out in zip(imgs, objs, output_paths):
            obj.append((img.height, img.width))
            with open(out, 'wb') as f:
                pickle.dump(obj, file=f)

    @torch.no_grad()
    def decompress_files(self, bits_paths):
        objs, sizes = [], []
        for p in bits_paths:
            with open(p, 'rb') as f:
                obj = pickle.load(file=f)
            sizes.append(obj.pop())
            objs.append(obj)
        out = self.decompress_batch(objs)
        return [out[i:i + 1, :, :h, :w] for i, (h, w) in enumerate(sizes)]

    @torch.no_grad()
    @on_model_device
    def encode_trace(self, im, full=False, force_z=None):
        """Per-block symbols / indexes of the encode plan (parity tests); `full` / `force_z` as in the qarv model's encode_trace."""
        B, _, H, W = im.shape
        self._prepare()
        pl = self._plan('enc', B, H, W)
        pl.im.view(B, 3, H, W).copy_(im)
        if full or force_z is not None:
            return self._trace_blocks(pl, B, force_z)
        pl.run()
        pl.fetch_status()
        torch.cuda.current_stream(pl.device).synchronize()
        pl.raise_if_flagged(where='(encode trace)')
        sym, idx = pl.sym_all.cpu().numpy(), pl.idx_all.cpu().numpy()
        return [dict(symbols=sym[o:o + B * z * hw].reshape(B, z, hw).copy(), indexes=idx[o:o + B * z * hw].reshape(B, z, hw).copy())
                for o, (z, hw) in zip(pl.sym_off, pl.lat_shapes)]

    @torch.no_grad()
    @on_model_device
    def cond_sample(self, latents, nhw_repeat=None, temprature=1.0, paint_box=None):
        """Decoder output for GIVEN latents (reference qresvae/model.py:591-603 with every latent supplied: forward_with_latents
        :403-417 uses them verbatim): latents[i] is a (B, z_i, h_i, w_i) tensor, e.g. what the encoder quantised.  Sampling the
        missing latents of a partial list (temprature, paint_box) belongs to the reference's generation demos, not to the
        compress / decompress path, and is not offered."""
        assert paint_box is None and all(z is not None for z in latents), 'cond_sample: every latent must be given'
        B, _, nH, nW = latents[0].shape
        self._prepare()
        pl = self._plan('dec', B, nH * 64, nW * 64)
        assert len(latents) == len(pl.lat_shapes)
        lo = 0
        for li, cut in enumerate(pl.cuts[:len(pl.lat_shapes)]):
            pl.run(lo, cut)
            z, hw = pl.lat_shapes[li]
            ld, M = pl.zhat_ld[li], B * hw
            assert tuple(latents[li].shape[:2]) == (B, z) and latents[li][0, 0].numel() == hw, f'latent {li}: {tuple(latents[li].shape)}'
            zt = latents[li].to(pl.device, torch.float32).reshape(B, z, hw).permute(0, 2, 1).reshape(M, z)
            zh = pl.zhat_bufs[li][:M * ld].view(M, ld)
            zh.zero_()
            zh[:, :z].copy_(zt)
            lo = cut + 1                                 # the launch at `cut` is this block's dequantize: skipped
        assert not pl.lossless, 'cond_sample: lossy models only (the lossless output net codes pixels, not a latent)'
        pl.run(lo, None)
        pl.fetch_status()
        torch.cuda.current_stream(pl.device).synchronize()
        pl.raise_if_flagged(where='in cond_sample()')
        return pl.out.clone()
