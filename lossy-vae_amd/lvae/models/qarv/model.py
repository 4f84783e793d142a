"""QARV (variable-rate hierarchical VAE) inference codec on MI355X.

API surface of the reference's `VariableRateLossyVAE` (lvae/models/qarv/model.py:169-581) for the encode/decode
path: `compress_mode`, `compress`, `decompress`, `compress_file`, `decompress_file`, `default_lmb`, `lmb_range`,
`max_stride`, `num_latents`, nn.Module behaviour (`.to`, `.eval`, `.parameters`, `load_state_dict` with the
reference's key names).  Bitstreams use the reference's container (qarv/model.py:525-528,567).

Nothing here calls a PyTorch compute kernel on the hot path: the module tree below only OWNS parameters under the
reference's names; `_prepare()` repacks them once into NHWC/GEMM-friendly device arrays and `_EncPlan`/`_DecPlan`
record the whole network as native HIP launches (lvae/engine.py).  Extension over the reference: `compress_batch` /
`decompress_batch` code B images per call (reference: batch 1 only, model.py:521) -- the GPU part runs batched, the
rANS streams of the B images x 9 latent blocks are coded by parallel host threads.
"""
import ctypes
import math
import os
import struct
import time

import numpy as np
import torch
import torch.nn as nn

from ... import _native
from ...engine import Plan, ptr
from ...utils import coding
from ..base import CodecBase, PREC_CODE, on_model_device
from ..entropy_coding import DiscretizedGaussian, rans_decode_streams, rans_encode_streams

EMBED_DIM = 256
# `model.side_streams` (default on since round 5): encode plans up to this many pixels per launch run posterior0 and the prior heads on a
# side stream (single images / small batches: the GPU is far from full and every launch of a branch is latency on the critical path:
# 0.3 ms of a single-image encode); larger batches fill the chip anyway.  Rounds 2-4 kept it opt-in: round 2 had seen run-to-run
# different bitstreams with kernels of two streams sharing CUs -- since traced, at ISA level, to a packed-FMA operand form
# (`op_sel` on src1: tools/ubench/pk_opsel_probe.hip, profiles/r04_ubench_pk_opsel_erratum_probe.txt) that the depthwise kernel no
# longer uses; the two pipeline groups of every batched call share CUs the same way.  Results do not depend on the option (same
# kernels, same inputs): tests/test_gpu_model.py::test_encode_is_stable_under_stream_concurrency.
SIDE_STREAM_MAX_PIXELS = 2 * 512 * 768


# ----------------------------------------------------------------------------------------------- parameter holders
class _Marker(nn.Module):
    """Parameter-less placeholder keeping list indices aligned with the reference (SetKey / CompresionStopFlag,
    lvae/models/common.py:48-66)."""
    def __init__(self, kind, key=None):
        super().__init__()
        self.kind, self.key = kind, key


class _MlpParams(nn.Module):
    def __init__(self, dim, hidden, out_dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, out_dim)


class CNXParams(nn.Module):
    """Parameters of one ConvNeXtBlockAdaLN (lvae/models/common.py:110-140): conv_dw, embedding_layer.1, mlp.fc1/fc2,
    gamma (1,C,1,1) initialised to 1e-6."""
    kind = 'cnx'

    def __init__(self, dim, kernel_size=7, mlp_ratio=2, embed_dim=EMBED_DIM):
        super().__init__()
        self.dim, self.kernel_size, self.hidden = dim, kernel_size, int(mlp_ratio * dim)
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, groups=dim)
        self.embedding_layer = nn.Sequential(nn.Identity(), nn.Linear(embed_dim, 2 * dim), nn.Identity())
        self.mlp = _MlpParams(dim, self.hidden, dim)
        self.gamma = nn.Parameter(torch.full(size=(1, dim, 1, 1), fill_value=1e-6))


def _conv(cin, cout, k, stride=1, padding=0):
    c = nn.Conv2d(cin, cout, k, stride, padding)
    c.bias.data.mul_(0.0)          # get_conv(zero_bias=True), common.py:8-14
    return c


class DownParams(nn.Conv2d):
    """patch_downsample (common.py:29-30): conv kernel=stride=rate."""
    kind = 'down'

    def __init__(self, cin, cout, rate):
        super().__init__(cin, cout, rate, rate, 0)
        self.bias.data.mul_(0.0)
        self.rate = rate


class UpParams(nn.Sequential):
    """patch_upsample (common.py:33-38): conv1x1 to cout*rate^2 channels + PixelShuffle(rate)."""
    kind = 'up'

    def __init__(self, cin, cout, rate):
        super().__init__(_conv(cin, cout * rate * rate, 1), nn.Identity())
        self.cin, self.cout, self.rate = cin, cout, rate


class VRLVParams(nn.Module):
    """Parameters of one VRLVBlockBase (qarv/model.py:19-42)."""
    kind = 'vrlv'

    def __init__(self, width, zdim, enc_key, enc_width, kernel_size=7, mlp_ratio=2):
        super().__init__()
        self.width, self.zdim, self.enc_key, self.enc_width, self.kernel_size = width, zdim, enc_key, enc_width, kernel_size
        self.resnet_front = CNXParams(width, kernel_size, mlp_ratio)
        self.resnet_end = CNXParams(width, kernel_size, mlp_ratio)
        self.posterior0 = CNXParams(enc_width, kernel_size)
        self.posterior1 = CNXParams(width, kernel_size)
        self.posterior2 = CNXParams(width, kernel_size)
        self.post_merge = _conv(width + enc_width, width, 1)
        self.posterior = _conv(width, zdim, 3, 1, 1)
        self.z_proj = _conv(zdim, width, 1)
        self.prior = _conv(width, zdim * 2, 1)
        self.discrete_gaussian = DiscretizedGaussian(cdf_form='erf')
        self.is_latent_block = True


class _Encoder(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.enc_blocks = nn.ModuleList(blocks)


# ----------------------------------------------------------------------------------------------- launch plans
class _Packed:
    """Device-resident, kernel-friendly copies of the weights (built once per device / weight version)."""

    def __init__(self, model, device):
        self.t = {}
        self.adaln_off = {}
        dev = device
        f32 = dict(device=dev, dtype=torch.float32)
        ws, bs = [], []
        total = 0

        def put(name, t):
            self.t[name] = t.detach().to(**f32).contiguous()

        def cnx(p, m):
            nonlocal total
            C, k = m.dim, m.kernel_size
            put(p + '.dw_w', m.conv_dw.weight.reshape(C, k * k).t())            # [k*k][C]
            put(p + '.dw_b', m.conv_dw.bias)
            put(p + '.fc1_w', m.mlp.fc1.weight); put(p + '.fc1_b', m.mlp.fc1.bias)
            put(p + '.fc2_w', m.mlp.fc2.weight); put(p + '.fc2_b', m.mlp.fc2.bias)
            put(p + '.gamma', m.gamma.reshape(C))
            lin = m.embedding_layer[1]
            b = lin.bias.detach().clone().float()
            b[C:] += 1.0                       # second half is `scale`; the kernel consumes (1 + scale) (common.py:151-152)
            ws.append(lin.weight.detach().float()); bs.append(b)
            self.adaln_off[p] = total
            total += 2 * C

        for i, m in enumerate(model.encoder.enc_blocks):
            p = f'encoder.enc_blocks.{i}'
            if m.kind == 'cnx':
                cnx(p, m)
            elif m.kind == 'down' and m.rate == 4:
                put(p + '.w', m.weight.reshape(m.out_channels, -1).t())           # [48][Cout], k=(ci*4+i)*4+j
                put(p + '.b', m.bias)
            elif m.kind == 'down':
                put(p + '.w', m.weight.permute(0, 2, 3, 1).reshape(m.out_channels, -1))   # [Cout][(i,j,ci)]
                put(p + '.b', m.bias)
        for i, m in enumerate(model.dec_blocks):
            p = f'dec_blocks.{i}'
            if m.kind == 'cnx':
                cnx(p, m)
            elif m.kind == 'up':
                w = m[0].weight.reshape(m.cout * m.rate ** 2, m.cin)
                b = m[0].bias
                if m.cout > 3:      # NHWC pixel-shuffle store wants columns ordered (i, j, c)
                    r2 = m.rate ** 2
                    w = w.reshape(m.cout, r2, m.cin).permute(1, 0, 2).reshape(r2 * m.cout, m.cin)
                    b = b.reshape(m.cout, r2).t().reshape(-1)
                put(p + '.w', w); put(p + '.b', b)
            elif m.kind == 'vrlv':
                for sub in ('resnet_front', 'resnet_end', 'posterior0', 'posterior1', 'posterior2'):
                    cnx(f'{p}.{sub}', getattr(m, sub))
                put(p + '.post_merge.w', m.post_merge.weight.reshape(m.width, -1)); put(p + '.post_merge.b', m.post_merge.bias)
                put(p + '.posterior.w', m.posterior.weight.permute(0, 2, 3, 1).reshape(m.zdim, -1))
                put(p + '.posterior.b', m.posterior.bias)
                put(p + '.z_proj.w', m.z_proj.weight.reshape(m.width, m.zdim)); put(p + '.z_proj.b', m.z_proj.bias)
                put(p + '.prior.w', m.prior.weight.reshape(2 * m.zdim, m.width)); put(p + '.prior.b', m.prior.bias)
        put('bias', model.bias.reshape(-1))
        put('lmb.0.w', model.lmb_embedding[0].weight); put('lmb.0.b', model.lmb_embedding[0].bias)
        put('lmb.2.w', model.lmb_embedding[2].weight); put('lmb.2.b', model.lmb_embedding[2].bias)
        put('adaln.w', torch.cat(ws, 0)); put('adaln.b', torch.cat(bs, 0))
        self.adaln_total = total
        self.adaln = torch.zeros(total, **f32)          # per-lambda (shift | 1+scale) vectors of all blocks
        self.emb_in = torch.zeros(EMBED_DIM, **f32)
        self.emb_h = torch.zeros(EMBED_DIM, **f32)
        self.emb = torch.zeros(EMBED_DIM, **f32)
        self.scale_table = model._dg().scale_table.detach().to(**f32).contiguous()
        self.scale_bound = float(model._dg().lower_bound_scale.bound.item())

    def p(self, name):
        return self.t[name].data_ptr()

    def bf16_map(self, mode):
        """Built once per mode, under a lock: plans are recorded concurrently by the pipeline-group threads, and a second
        builder would free the first one's bf16 copies while its plan still points at them."""
        from ..base import (bf16_weight_map, bf16x3_weight_map, f16x2_weight_map, f16x2k32_weight_map, mxfp8_weight_map,
                            mxfp8q8_weight_map, _W16_LOCK)
        with _W16_LOCK:
            if not hasattr(self, '_w16'):
                self._w16 = {}
            if mode not in self._w16:
                self._w16[mode] = {'bf16': bf16_weight_map, 'bf16x3': bf16x3_weight_map, 'f16x2': f16x2_weight_map,
                                   'f16x2k32': f16x2k32_weight_map, 'fp8': mxfp8_weight_map, 'mxfp8q8': mxfp8q8_weight_map}[mode](self.t)
        return self._w16[mode][0]


class _NetPlan(Plan):
    """Shared recording helpers for the encode and decode plans."""

    def __init__(self, model, pk, B):
        super().__init__(pk.adaln.device)
        self.model, self.pk, self.B = model, pk, B
        self.prec = PREC_CODE[model._prec]
        self.prec_name = model._prec
        self.w16 = pk.bf16_map(model._prec) if self.prec else None
        self.w16_x3 = pk.bf16_map('bf16x3') if self.prec == 4 else None
        self.w16_k32 = pk.bf16_map('f16x2k32') if self.prec == 4 else None
        self.w16_q8 = pk.bf16_map('mxfp8q8') if self.prec == 3 else None
        self.lp = self.prec == 3                # reduced precision (BASELINE config 5): feature maps stored as bf16, MX-fp8 GEMMs
        self.adt = torch.bfloat16 if self.lp else torch.float32
        self.dwln = self.lib.lvae_dwconv_ln_bf16 if self.lp else self.lib.lvae_dwconv_ln_f32
        self.sym_off, self.idx_off = [], []     # per latent block element offsets into sym_all / idx_all
        self.pm_bufs = []                       # per latent block prior means [M][z] (NHWC rows)
        self.qcuts = []                         # encode plans: op index right after each block's quantize launch
        self.prm_ptrs, self.zhat_ptrs, self.zhat_bufs = [], [], []  # per latent block: raw prior conv output / latent buffer (scratch may be re-grown)
        self.prm_bufs, self.qm_bufs, self.zhat_ld = [], [], []      # test access (CodecBase._trace_blocks): tensors behind those launches
        self.lat_shapes = []                    # (z, HW)

    def scratch(self, M, C, hid):
        y = self.buf(self.sname('y'), M * C, self.adt)
        h = self.buf(self.sname('hid'), M * hid, self.adt)
        return y, h

    def cnx(self, p, m, x, out, H, W):
        """ConvNeXtBlockAdaLN (common.py:142-161) = dwconv+LN+AdaLN kernel, fc1+GELU GEMM, fc2+gamma+residual GEMM."""
        pk, lib = self.pk, self.lib
        C, k, hid = m.dim, m.kernel_size, m.hidden
        M = self.B * H * W
        y, h = self.scratch(M, C, hid)
        off = pk.adaln_off[p]
        # f16x2 plans: y and the hidden map have one consumer each (fc1 / fc2), so their producers store them pre-split (hi / lo' fp16
        # planes, 4 bytes per element like fp32) and the two GEMMs stream both operands by LDS-DMA with no conversion in the main loop
        # (reduced-precision plans: the same idea with MX-fp8 -- the producers quantise, csrc/gemm_q8.hip streams)
        # (small maps: the split-K layers -- pre-split + serial split-K where the batch makes that the faster form, same bits: engine.mlp_pipeline)
        if self.mlp_fused_ok(C, hid, k, M=M, rows_per_image=H * W):
            # C = 128 / hidden = 192 (the decoder's stride-4 blocks): fc1 -> GELU -> fc2 as one launch, the hidden tile never leaves the CU
            self.add(lib.lvae_dwconv_ln_h2, (x, pk.p(p + '.dw_w'), pk.p(p + '.dw_b'), None, None, ptr(pk.adaln, off), ptr(pk.adaln, off + C),
                                             y.data_ptr(), self.B, H, W, C, k), p + '.dwln')
            self.mlp_fused(y=y.data_ptr(), M=M, C=C, hid=hid, w1=pk.p(p + '.fc1_w'), b1=pk.p(p + '.fc1_b'), w2=pk.p(p + '.fc2_w'),
                           b2=pk.p(p + '.fc2_b'), gamma=pk.p(p + '.gamma'), res=x, out=out, label=p + '.mlp')
            return
        sk = self.mlp_sk_ok(C, hid, k, H * W, M)
        if sk is not None:
            # stride-32 / 64 maps (both GEMMs split-K): fc1 -> GELU -> fc2's partial sums as one launch, then the reduce launch (csrc/mlp_sk.hip)
            self.add(lib.lvae_dwconv_ln_h2, (x, pk.p(p + '.dw_w'), pk.p(p + '.dw_b'), None, None, ptr(pk.adaln, off), ptr(pk.adaln, off + C),
                                             y.data_ptr(), self.B, H, W, C, k), p + '.dwln')
            self.mlp_sk(y=y.data_ptr(), M=M, C=C, hid=hid, S1=sk[0], S2=sk[1], w1=pk.p(p + '.fc1_w'), b1=pk.p(p + '.fc1_b'), w2=pk.p(p + '.fc2_w'),
                        b2=pk.p(p + '.fc2_b'), gamma=pk.p(p + '.gamma'), res=x, out=out, label=p + '.mlp')
            return
        if self.mlp_q8_ok(C, hid, k):
            pre1, pre2, S1, S2 = True, True, None, None
        else:
            pre1, pre2, S1, S2 = self.mlp_pipeline(C, hid, k, H * W)
        self.add((lib.lvae_dwconv_ln_q8 if self.lp else lib.lvae_dwconv_ln_h2) if pre1 else self.dwln, (x, pk.p(p + '.dw_w'), pk.p(p + '.dw_b'), None, None, ptr(pk.adaln, off),
                                                               ptr(pk.adaln, off + C), y.data_ptr(), self.B, H, W, C, k), p + '.dwln')
        self.gemm(A0=y.data_ptr(), K0=C, M=M, N=hid, Wt=pk.p(p + '.fc1_w'), bias=pk.p(p + '.fc1_b'), out=h.data_ptr(),
                  epi=_native.EPI_BIAS_GELU, a_h2=pre1, out_h2=pre2, ksplit=S1, label=p + '.fc1')
        self.gemm(A0=h.data_ptr(), K0=hid, M=M, N=C, Wt=pk.p(p + '.fc2_w'), bias=pk.p(p + '.fc2_b'),
                  gamma=pk.p(p + '.gamma'), res=x, ldres=C, out=out, epi=_native.EPI_GAMMA_RES, a_h2=pre2, ksplit=S2, label=p + '.fc2')

    def upsample(self, p, m, x, out, H, W):
        pk = self.pk
        M = self.B * H * W
        final = m.cout <= 3
        self.gemm(A0=x, K0=m.cin, M=M, N=m.cout * m.rate ** 2, Wt=pk.p(p + '.w'), bias=pk.p(p + '.b'), out=out,
                  store=_native.ST_IMAGE if final else _native.ST_SHUFFLE, r=m.rate, H=H, W=W, label=p + '.up')

    def prior(self, p, m, f, H, W, side_head=False):
        """transform_prior (qarv/model.py:44-54) + build_indexes (:106/:112). Returns pm buffer."""
        pk, lib, B = self.pk, self.lib, self.B
        M, z = B * H * W, m.zdim
        self.cnx(p + '.resnet_front', m.resnet_front, f, f, H, W)
        if side_head:          # encoder: the prior head is off the critical path until quantize -- it runs beside posterior1 / post_merge
            self.fork(p + '.fork_prior')
            self.side_begin()
        prm = self.buf('prm', M * 2 * z)
        # (split-K launches leave their reduce pass to the index kernel: one launch less per latent block on both chains, same bits)
        planes = self.gemm(A0=f, K0=m.width, M=M, N=2 * z, Wt=pk.p(p + '.prior.w'), bias=pk.p(p + '.prior.b'),
                           out=prm.data_ptr(), out_bf16=0, defer_reduce=(self.prec != 3 and self.DEFER_HEAD_REDUCE), label=p + '.prior')
        pm = self.new(M * z)
        self.pm_bufs.append(pm)
        self.prm_ptrs.append(prm.data_ptr())
        self.prm_bufs.append(prm)
        ioff = sum(s[0] * s[1] for s in self.lat_shapes) * B
        self.lat_shapes.append((z, H * W))
        self.idx_off.append(ioff)
        if planes is not None:
            self.add(lib.lvae_prior_index_sk_f32, (planes[0], planes[1], pk.p(p + '.prior.b'), prm.data_ptr(), pm.data_ptr(), ptr(self.idx_all, ioff),
                                                   pk.scale_table.data_ptr(), pk.scale_table.numel(), pk.scale_bound, B, H * W, z, self.status_ptr()),
                     p + '.prior_index')
        else:
            self.add(lib.lvae_prior_index_f32, (prm.data_ptr(), pm.data_ptr(), ptr(self.idx_all, ioff), pk.scale_table.data_ptr(),
                                                pk.scale_table.numel(), pk.scale_bound, B, H * W, z, self.status_ptr()), p + '.prior_index')
        self.side_end()
        return pm, ioff

    def fuse_and_end(self, p, m, f, zhat, H, W):
        """fuse_feature_and_z + resnet_end (qarv/model.py:72-75,117-118)."""
        pk = self.pk
        M = self.B * H * W
        self.gemm(A0=zhat, K0=m.zdim, M=M, N=m.width, Wt=pk.p(p + '.z_proj.w'), bias=pk.p(p + '.z_proj.b'), res=f,
                  ldres=m.width, out=f, epi=_native.EPI_RES, a_bf16=0, label=p + '.z_proj')
        self.cnx(p + '.resnet_end', m.resnet_end, f, f, H, W)

    def alloc_latent_io(self, nH, nW):
        B = self.B
        total, s = 0, 1
        # latent resolution per block follows the top-down path: starts at (nH,nW), doubles at each upsample
        for m in self.model.dec_blocks:
            if m.kind == 'vrlv':
                total += m.zdim * nH * s * nW * s
            elif m.kind == 'up':
                s *= m.rate
            elif m.kind == 'stop':
                break
        self.n_sym = total * B
        self.sym_all = self.new(self.n_sym, torch.int32)
        self.idx_all = self.new(self.n_sym, torch.uint8)
        self.sym_host = torch.empty(self.n_sym, dtype=torch.int32).pin_memory()
        self.idx_host = torch.empty(self.n_sym, dtype=torch.uint8).pin_memory()
        self.sym_np, self.idx_np = self.sym_host.numpy(), self.idx_host.numpy()


class _EncPlan(_NetPlan):
    """forward_end2end(mode='compress') (qarv/model.py:294-315) for B images of size HxW."""

    def __init__(self, model, pk, B, H, W, with_bits=False):
        super().__init__(model, pk, B)
        lib = self.lib
        self.im = self.new(B * 3 * H * W)
        if getattr(model, 'side_streams', False):
            self.enable_side_stream()
        # small plans: every block's posterior0 and prior head beside the main branch (pure launch latency there)
        small_side = self.side_stream is not None and B * H * W <= SIDE_STREAM_MAX_PIXELS
        # every plan: posterior0 of the stride-8 / 16 latent blocks -- a ConvNeXt block on the ENCODER feature alone (qarv/model.py:58-59),
        # the only large launches of the encode that do not sit on its dependency chain -- is hoisted to the point where the bottom-up
        # path leaves stride 16 and runs on the side stream under the stride-32 / 64 stages of both paths, whose launches (M = 96 ... 384
        # rows per image) leave the chip almost empty (round 5; same kernels, same inputs, same bits)
        hoisted = {}                                    # dec_blocks index -> buffer holding posterior0's output
        self.alloc_latent_io(H // 64, W // 64)
        self.nats = self.new(model.num_latents * B, torch.float64) if with_bits else None   # [block][image] sum(-ln P)
        feats = {}
        tapped = set()
        h, w = H, W
        x = None
        for i, m in enumerate(model.encoder.enc_blocks):
            p = f'encoder.enc_blocks.{i}'
            if m.kind == 'down' and m.rate == 4:
                h, w = h // 4, w // 4
                x = self.new(B * h * w * m.out_channels, self.adt)
                self.add(lib.lvae_stem_bf16 if self.lp else lib.lvae_stem_f32, (self.im.data_ptr(), pk.p(p + '.w'), pk.p(p + '.b'), x.data_ptr(), B, H, W,
                                             m.out_channels, model.im_shift, model.im_scale, self.status_ptr()), p + '.stem')
                self.flops += 2 * B * h * w * m.out_channels * 48
            elif m.kind == 'down':
                if self.side_stream is not None and (h, w) == (H // 16, W // 16) and not hoisted and (not small_side or getattr(model, 'hoist_small', True)):
                    self._hoist_posterior0(model, feats, hoisted, H, W)
                h, w = h // 2, w // 2
                nx = self.new(B * h * w * m.out_channels, self.adt)
                self.gemm(A0=x.data_ptr(), K0=m.in_channels, M=B * h * w, N=m.out_channels, K=4 * m.in_channels,
                          Wt=pk.p(p + '.w'), bias=pk.p(p + '.b'), out=nx.data_ptr(), a_mode=_native.A_PATCH2, H=h, W=w,
                          label=p + '.down')
                x = nx
            elif m.kind == 'cnx':
                if x.data_ptr() in tapped:               # feature was tapped by SetKey: keep it, write elsewhere
                    nx = self.new(x.numel(), self.adt)
                    self.cnx(p, m, x.data_ptr(), nx.data_ptr(), h, w)
                    x = nx
                else:
                    self.cnx(p, m, x.data_ptr(), x.data_ptr(), h, w)
            elif m.kind == 'key':
                feats[m.key] = (x, h, w)
                tapped.add(x.data_ptr())
        # top-down path
        h, w = H // 64, W // 64
        width = model.dec_blocks[0].width
        f = self.new(B * h * w * width, self.adt)
        self.add(lib.lvae_bias_expand_bf16 if self.lp else lib.lvae_bias_expand_f32, (pk.p('bias'), f.data_ptr(), B * h * w, width), 'bias')
        for i, m in enumerate(model.dec_blocks):
            p = f'dec_blocks.{i}'
            if m.kind == 'vrlv':
                M, z = B * h * w, m.zdim
                ef, eh, ew = feats[m.enc_key]
                assert (eh, ew) == (h, w)
                g = self.buf('post_g', M * m.width, self.adt)
                mg = self.buf('post_m', M * m.width, self.adt)
                # posterior0 works on the ENCODER feature only (qarv/model.py:56-70): hoisted (above), or -- small plans -- on the side
                # stream beside resnet_front (and the prior head beside posterior1), or in line; side work is joined before post_merge
                if i in hoisted:
                    e = hoisted[i]
                else:
                    e = self.buf('post_e', M * m.enc_width, self.adt)
                    if small_side:
                        self.fork(p + '.fork_post0')
                        self.side_begin()
                    self.cnx(p + '.posterior0', m.posterior0, ef.data_ptr(), e.data_ptr(), h, w)
                    self.side_end()
                # (prior heads on the side stream in EVERY plan were measured too: +0.27 ms per encode at batch 8, profiles/r05_ab_encode_side_stream.txt)
                head_side = small_side
                pm, ioff = self.prior(p, m, f.data_ptr(), h, w, side_head=head_side)
                self.cnx(p + '.posterior1', m.posterior1, f.data_ptr(), g.data_ptr(), h, w)
                if head_side or i in hoisted:
                    self.join(p + '.join')
                self.gemm(A0=g.data_ptr(), K0=m.width, A1=e.data_ptr(), K1=m.enc_width, lda1=m.enc_width, M=M, N=m.width,
                          Wt=pk.p(p + '.post_merge.w'), bias=pk.p(p + '.post_merge.b'), out=mg.data_ptr(),
                          label=p + '.post_merge')
                self.cnx(p + '.posterior2', m.posterior2, mg.data_ptr(), mg.data_ptr(), h, w)
                qm = self.buf('qm', M * z)
                planes = self.gemm(A0=mg.data_ptr(), K0=m.width, M=M, N=z, K=9 * m.width, Wt=pk.p(p + '.posterior.w'),
                                   bias=pk.p(p + '.posterior.b'), out=qm.data_ptr(), a_mode=_native.A_CONV3, H=h, W=w, out_bf16=0,
                                   defer_reduce=(self.prec != 3 and self.DEFER_HEAD_REDUCE), label=p + '.posterior')
                zhat = self.buf('zhat', M * z)
                self.qm_bufs.append(qm); self.zhat_bufs.append(zhat); self.zhat_ld.append(z)
                self.sym_off.append(ioff)
                if planes:      # split-K planes of the posterior head: summed (slice order, + bias) by the quantize launch itself
                    self.add(lib.lvae_quantize_sk_f32, (planes[0], planes[1], pk.p(p + '.posterior.b'), qm.data_ptr(), pm.data_ptr(),
                                                        ptr(self.sym_all, ioff), zhat.data_ptr(), B, h * w, z, z, self.status_ptr()), p + '.quantize')
                else:
                    self.add(lib.lvae_quantize_f32, (qm.data_ptr(), pm.data_ptr(), ptr(self.sym_all, ioff), zhat.data_ptr(),
                                                     B, h * w, z, z, self.status_ptr()), p + '.quantize')
                self.qcuts.append(len(self.ops))        # this block's symbols and indexes are final from here on
                if with_bits:       # eval-mode likelihood of the quantised latent (qarv/model.py:95-96), prm still holds this block
                    li = len(self.sym_off) - 1
                    self.add(lib.lvae_gaussian_nll_f32, (self.bufs['prm'].data_ptr(), ptr(self.sym_all, ioff), ptr(self.nats, li * B),
                                                         pk.scale_bound, B, h * w, z, 0), p + '.nll')
                self.fuse_and_end(p, m, f.data_ptr(), zhat.data_ptr(), h, w)
            elif m.kind == 'cnx':
                self.cnx(p, m, f.data_ptr(), f.data_ptr(), h, w)
            elif m.kind == 'up':
                nf = self.new(B * h * w * m.rate ** 2 * m.cout, self.adt)
                self.upsample(p, m, f.data_ptr(), nf.data_ptr(), h, w)
                f = nf
                h, w = h * m.rate, w * m.rate
            elif m.kind == 'stop':
                break                                                     # qarv/model.py:310-312


    def _hoist_posterior0(self, model, feats, hoisted, H, W):
        """Record posterior0 of every latent block whose encoder feature is already there (strides 8 and 16) on the side stream,
        the blocks the top-down path reaches first (stride 16) first."""
        todo, s = [], 1
        for i, m in enumerate(model.dec_blocks):
            if m.kind == 'vrlv' and m.enc_key in feats:
                todo.append((i, m))
            elif m.kind == 'stop':
                break
        if not todo:
            return
        self.fork('hoist.fork')
        self.side_begin()
        for i, m in todo:                               # dec_blocks order = the order the top-down path needs them in
            ef, eh, ew = feats[m.enc_key]
            e = self.new(self.B * eh * ew * m.enc_width, self.adt)
            self.cnx(f'dec_blocks.{i}.posterior0', m.posterior0, ef.data_ptr(), e.data_ptr(), eh, ew)
            hoisted[i] = e
        self.side_end()


class _DecPlan(_NetPlan):
    """decompress() (qarv/model.py:531-557): 9 GPU segments separated by host rANS decodes."""

    def __init__(self, model, pk, B, nH, nW):
        super().__init__(model, pk, B)
        lib = self.lib
        self.alloc_latent_io_full(nH, nW)
        h, w = nH, nW
        width = model.dec_blocks[0].width
        f = self.new(B * h * w * width, self.adt)
        self.add(lib.lvae_bias_expand_bf16 if self.lp else lib.lvae_bias_expand_f32, (pk.p('bias'), f.data_ptr(), B * h * w, width), 'bias')
        self.cuts = []        # op index after each prior_index (host decode happens there)
        self.lat_hw = []      # (h, w) of each latent block
        self.out = None
        for i, m in enumerate(model.dec_blocks):
            p = f'dec_blocks.{i}'
            if m.kind == 'vrlv':
                M, z = B * h * w, m.zdim
                pm, ioff = self.prior(p, m, f.data_ptr(), h, w)
                self.cuts.append(len(self.ops))
                self.sym_off.append(ioff)
                self.lat_hw.append((h, w))
                zhat = self.buf('zhat', M * z)
                self.zhat_ptrs.append(zhat.data_ptr())
                self.zhat_bufs.append(zhat)
                self.add(lib.lvae_dequantize_f32, (ptr(self.sym_all, ioff), pm.data_ptr(), zhat.data_ptr(), B, h * w, z, z),
                         p + '.dequantize')
                self.fuse_and_end(p, m, f.data_ptr(), zhat.data_ptr(), h, w)
            elif m.kind == 'cnx':
                self.cnx(p, m, f.data_ptr(), f.data_ptr(), h, w)
            elif m.kind == 'up':
                final = m.cout <= 3
                nf = self.new(B * h * w * m.rate ** 2 * m.cout, torch.float32 if final else self.adt)
                self.upsample(p, m, f.data_ptr(), nf.data_ptr(), h, w)
                f = nf
                h, w = h * m.rate, w * m.rate
                if final:
                    self.out = nf.view(B, m.cout, h, w)
        assert self.out is not None

    def alloc_latent_io_full(self, nH, nW):
        self.alloc_latent_io(nH, nW)


# ----------------------------------------------------------------------------------------------- the model
class VariableRateLossyVAE(CodecBase):
    log2_e = math.log2(math.e)
    MAX_LMB = 8192

    def __init__(self, config: dict):
        super().__init__()
        self.encoder = _Encoder(config.pop('enc_blocks'))
        self.dec_blocks = nn.ModuleList(config.pop('dec_blocks'))
        width = self.dec_blocks[0].width
        self.bias = nn.Parameter(torch.zeros(1, width, 1, 1))
        self.num_latents = len([b for b in self.dec_blocks if getattr(b, 'is_latent_block', False)])

        _low, _high = config['lmb_range']
        self.lmb_range = (float(_low), float(_high))
        self.default_lmb = self.lmb_range[1]
        self.lmb_embed_dim = config['lmb_embed_dim']
        self.lmb_embedding = nn.Sequential(
            nn.Linear(self.lmb_embed_dim[0], self.lmb_embed_dim[1]), nn.GELU(),
            nn.Linear(self.lmb_embed_dim[1], self.lmb_embed_dim[1]))
        self._sin_period = config['sin_period']

        self.im_shift = float(config['im_shift'])
        self.im_scale = float(config['im_scale'])
        self.max_stride = config['max_stride']
        self.register_buffer('_dummy', torch.zeros(1), persistent=False)
        self.compressing = False
        self._init_codec_base()
        self._packed = None
        self._packed_key = None
        self._plans = {}
        self._cur_lmb = None
        self.timing = {} if os.environ.get('LVAE_TIMING') else None      # host-side phase timers (debug)
        # independent encoder branches on a second HIP stream (see SIDE_STREAM_MAX_PIXELS, _EncPlan); LVAE_SIDE_STREAMS=0: A/B switch, same bits
        self.side_streams = os.environ.get('LVAE_SIDE_STREAMS', '1') == '1'

    # ---- helpers
    def _dg(self) -> DiscretizedGaussian:
        for b in self.dec_blocks:
            if getattr(b, 'is_latent_block', False):
                return b.discrete_gaussian
        raise RuntimeError('no latent block')

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def _invalidate(self):
        self._packed, self._plans, self._cur_lmb = None, {}, None

    def _prepare(self):
        dev = self._dummy.device
        if self._packed is None or self._packed.adaln.device != dev:
            if dev.type != 'cuda':
                raise RuntimeError('lvae (MI355X build): compress/decompress run on the GPU only; move the model with '
                                   '.to("cuda") -- there is deliberately no CPU fallback')
            _native.lib()
            with torch.no_grad():
                self._packed = _Packed(self, dev)
            self._plans, self._cur_lmb = {}, None
        return self._packed

    def _set_lmb(self, lmb):
        """_get_lmb_embedding (qarv/model.py:266-287) + every block's AdaLN embedding_layer (common.py:150-151), once
        per lambda: sinusoidal features on the host (128 cos + 128 sin), then three GEMV launches."""
        pk = self._prepare()
        lmb = float(np.float32(lmb))
        if self._cur_lmb == lmb:
            return
        if torch.cuda.current_device() != pk.adaln.device.index:   # raw launches below: make the model's GPU the current one
            with torch.cuda.device(pk.adaln.device):
                return self._set_lmb(lmb)
        s = np.log(np.float32(lmb)) * np.float32(self._sin_period) / np.float32(math.log(self.MAX_LMB))
        dim = self.lmb_embed_dim[0]
        expo = np.linspace(0, 1, dim // 2, dtype=np.float32)
        freqs = np.power(np.float32(self._sin_period), -expo).astype(np.float32)
        args = (np.float32(s) * freqs).astype(np.float32)
        e = np.concatenate([np.cos(args), np.sin(args)]).astype(np.float32)
        pk.emb_in.copy_(torch.from_numpy(e))
        lib = _native.lib()
        st = torch.cuda.current_stream(pk.adaln.device).cuda_stream
        _native.check(lib.lvae_gemv_f32(pk.p('lmb.0.w'), pk.p('lmb.0.b'), pk.emb_in.data_ptr(), pk.emb_h.data_ptr(),
                                        self.lmb_embed_dim[1], dim, 0, 1, st), 'gemv lmb.0')
        _native.check(lib.lvae_gemv_f32(pk.p('lmb.2.w'), pk.p('lmb.2.b'), pk.emb_h.data_ptr(), pk.emb.data_ptr(),
                                        self.lmb_embed_dim[1], self.lmb_embed_dim[1], 0, 0, st), 'gemv lmb.2')
        _native.check(lib.lvae_gemv_f32(pk.p('adaln.w'), pk.p('adaln.b'), pk.emb.data_ptr(), pk.adaln.data_ptr(),
                                        pk.adaln_total, self.lmb_embed_dim[1], 1, 0, st), 'gemv adaln')
        self._cur_lmb = lmb

    def _plan(self, kind, B, a, b, group=0):
        key = (kind, B, a, b, group, bool(getattr(self, 'side_streams', False)) and kind != 'dec', self._prec)
        pl = self._plans.get(key)
        if pl is None:
            pk = self._prepare()
            if kind == 'enc':
                pl = _EncPlan(self, pk, B, a, b)
            elif kind == 'encb':
                pl = _EncPlan(self, pk, B, a, b, with_bits=True)
            else:
                pl = _DecPlan(self, pk, B, a, b)
            self._plans[key] = pl
        return pl

    # ---- reference API
    def compress_mode(self, mode=True):
        """qarv/model.py:509-514: (re)build the CDF tables of every latent block."""
        if mode:
            first = None
            for block in self.dec_blocks:
                if getattr(block, 'is_latent_block', False):
                    dg = block.discrete_gaussian
                    if first is None:
                        dg.update()
                        first = dg
                    else:       # all blocks share one scale table: identical rows, build once
                        dg._quantized_cdf, dg._offset, dg._cdf_length = first._quantized_cdf, first._offset, first._cdf_length
                        dg._host = None
            self._log_precision()
        self.compressing = mode

    @torch.no_grad()
    @on_model_device
    def compress_batch(self, im, lmb=None):
        """Encode a (B,3,H,W) batch -> list of B byte strings (each identical to `compress(im[b:b+1])`)."""
        lmb = lmb or self.default_lmb
        assert im.dim() == 4 and im.shape[1] == 3 and not im.requires_grad
        B, _, H, W = im.shape
        assert (H % self.max_stride == 0) and (W % self.max_stride == 0), f'{im.shape=}'
        self._prepare()
        self._set_lmb(lmb)
        tables = self._dg().host_tables()
        header = struct.pack('f', lmb) + struct.pack('3H', 1, H // self.max_stride, W // self.max_stride)
        groups = self._groups(B, 'enc')
        nthreads = self._coder_threads_per_group(len(groups))
        T = self.timing

        def encode_group(g, start, n, stream):
            pl = self._plan('enc', n, H, W, g)
            t0 = time.time()
            pl.im.view(n, 3, H, W).copy_(im[start:start + n])
            if self.native_group_loops:
                # the loop below as ONE foreign call (csrc/plan_runtime.cpp::lvae_encode_blocks): no interpreter between the launches,
                # the event waits and the coder calls, and no interpreter lock shared with the other group's thread
                per_block = self._encode_group_native(pl, pl.qcuts, pl.sym_off, n, tables, nthreads, stream, T)
                nl = len(pl.lat_shapes)
                assert nl == self.num_latents
                strings = [per_block[li][b] for b in range(n) for li in range(nl)]
                res = [header + coding.pack_byte_strings(strings[b * nl:(b + 1) * nl]) for b in range(n)]
                if T is not None:
                    T['enc_group_total'] = T.get('enc_group_total', 0) + time.time() - t0
                return res
            # Progressive hand-over: after each latent block's quantize launch, its symbols / indexes are copied to pinned host
            # memory and an event is recorded; the host then entropy-codes block i while the GPU is still computing blocks > i
            # (only the last block's streams are coded after the GPU has finished).
            lo, evs = 0, []
            for li, cut in enumerate(pl.qcuts):
                pl.run(lo, cut, stream=stream.cuda_stream)
                lo = cut
                z, hw = pl.lat_shapes[li]
                o, cnt = pl.sym_off[li], n * z * hw
                pl.sym_host[o:o + cnt].copy_(pl.sym_all[o:o + cnt], non_blocking=True)
                pl.idx_host[o:o + cnt].copy_(pl.idx_all[o:o + cnt], non_blocking=True)
                if li == len(pl.qcuts) - 1:
                    pl.fetch_status()
                ev = torch.cuda.Event()
                ev.record(stream)
                evs.append(ev)
            # (the launches after the last quantize -- z_proj / resnet_end of the last latent block, which the reference also runs
            # before it meets CompresionStopFlag -- do not influence the bitstream and are skipped)
            t1 = time.time()
            nl = len(pl.lat_shapes)
            per_block, t_wait = [], 0.0
            evs[-1].synchronize()       # (debug loop: the status word is read once, behind the last block -- no progressive hand-over)
            pl.raise_if_flagged(where='while encoding')
            for li, ev in enumerate(evs):
                tw = time.time()
                ev.synchronize()
                t_wait += time.time() - tw
                z, hw = pl.lat_shapes[li]
                o = pl.sym_off[li]
                sv = [pl.sym_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                iv = [pl.idx_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                per_block.append(rans_encode_streams(tables, sv, iv, nthreads))
            tw = time.time()
            stream.synchronize()
            t2 = t1 + t_wait + (time.time() - tw)
            strings = [per_block[li][b] for b in range(n) for li in range(nl)]
            t_gpu_done = time.time()
            assert nl == self.num_latents
            res = [header + coding.pack_byte_strings(strings[b * nl:(b + 1) * nl]) for b in range(n)]
            if T is not None:
                t3 = time.time()
                T['enc_launch'] = T.get('enc_launch', 0) + t1 - t0
                T['enc_gpu_wait'] = T.get('enc_gpu_wait', 0) + t2 - t1
                T['enc_rans'] = T.get('enc_rans', 0) + t3 - t2
                T['enc_last_wait'] = T.get('enc_last_wait', 0) + t_gpu_done - tw          # final stream.synchronize()
                T['enc_after_gpu'] = T.get('enc_after_gpu', 0) + t3 - t_gpu_done           # container packing after the GPU is done
                T['enc_group_total'] = T.get('enc_group_total', 0) + t3 - t0
            return res

        out = []
        for part in self._run_groups(encode_group, groups):
            out += part
        return out

    @torch.no_grad()
    def compress(self, im, lmb=None):
        """qarv/model.py:516-529 (single image)."""
        assert im.shape[0] == 1, f'Right now only support a single image, got {im.shape=}'
        return self.compress_batch(im, lmb)[0]

    @torch.no_grad()
    @on_model_device
    def decompress_batch(self, strings):
        """Decode a list of byte strings that share lambda and latent shape -> (B,3,H,W) tensor in [0,1]."""
        t_entry = time.time()
        B = len(strings)
        heads = [struct.unpack('f', s[:4]) + struct.unpack('3H', s[4:10]) for s in strings]
        lmb, nB, nH, nW = heads[0]
        assert nB == 1 and all(h == heads[0] for h in heads), 'batch must share lambda and shape'
        if not all(isinstance(s, bytes) for s in strings):
            strings = [bytes(s) for s in strings]
        lv = None if self.native_group_loops else [coding.unpack_byte_string(s[10:]) for s in strings]

        def stream_views(start, n, nb):
            """[image][block] -> (container, offset, length): the container's payloads (utils/coding.py: 'B' count, count x 'I' lengths,
            payloads) as views -- the native decode loop reads them in place, nothing is sliced out (2.4 MB of copies per batch of 8)."""
            out = []
            for b in range(n):
                s = strings[start + b]
                num = s[10]
                lengths = struct.unpack_from(f'{num}I', s, 11)
                o = 11 + 4 * num
                assert num == nb, f'expected {nb} strings per image'
                assert sum(lengths) == len(s) - o, f'{sum(lengths)=} should equal to {len(s) - o=}'
                v = []
                for ln in lengths:
                    v.append((s, o, ln))
                    o += ln
                out.append(v)
            return out
        t_a = time.time()
        self._prepare()
        self._set_lmb(lmb)
        tables = self._dg().host_tables()
        groups = self._groups(B, 'dec')
        nthreads = self._coder_threads_per_group(len(groups))
        T = self.timing
        out = torch.empty(B, 3, nH * self.max_stride, nW * self.max_stride, device=self._dummy.device)
        t_b = time.time()
        if T is not None:
            T['dec_head_parse'] = T.get('dec_head_parse', 0) + t_a - t_entry
            T['dec_head_setup'] = T.get('dec_head_setup', 0) + t_b - t_a

        def decode_group(g, start, n, stream):
            if T is not None:
                T['dec_head_thread'] = T.get('dec_head_thread', 0) + time.time() - t_b          # submit -> the group's thread runs
            pl = self._plan('dec', n, nH, nW, g)
            lo = 0
            if T is not None:
                T['dec_head'] = T.get('dec_head', 0) + time.time() - t_entry                    # entry -> this group's first launch
            if self.native_group_loops:
                # the loop below as ONE foreign call (csrc/plan_runtime.cpp::lvae_decode_blocks)
                self._decode_group_native(pl, pl.cuts, pl.idx_off, n, lambda: stream_views(start, n, len(pl.cuts)), tables, nthreads, stream, T)
                out[start:start + n].copy_(pl.out, non_blocking=True)
                return None
            assert all(len(lv[start + b]) == len(pl.cuts) for b in range(n)), f'expected {len(pl.cuts)} strings per image'
            for li, cut in enumerate(pl.cuts):
                t0 = time.time()
                pl.run(lo, cut, stream=stream.cuda_stream)
                lo = cut
                z, hw = pl.lat_shapes[li]
                o, cnt = pl.idx_off[li], n * z * hw
                pl.idx_host[o:o + cnt].copy_(pl.idx_all[o:o + cnt], non_blocking=True)
                stream.synchronize()
                t1 = time.time()
                iv = [pl.idx_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                sv = [pl.sym_np[o + b * z * hw:o + (b + 1) * z * hw] for b in range(n)]
                rans_decode_streams(tables, [lv[start + b][li] for b in range(n)], iv, sv, nthreads)
                pl.sym_all[o:o + cnt].copy_(pl.sym_host[o:o + cnt], non_blocking=True)
                if T is not None:
                    t2 = time.time()
                    T['dec_gpu_seg'] = T.get('dec_gpu_seg', 0) + t1 - t0
                    T['dec_rans'] = T.get('dec_rans', 0) + t2 - t1
            pl.run(lo, None, stream=stream.cuda_stream)
            pl.fetch_status()                               # read by _check_decoded() after the groups have finished
            out[start:start + n].copy_(pl.out, non_blocking=True)
            return None

        if T is not None:
            t_g = time.time()
        self._run_groups(decode_group, groups)
        self._check_decoded(groups, lambda g, n: self._plan('dec', n, nH, nW, g))
        if T is not None:
            T['dec_groups_total'] = T.get('dec_groups_total', 0) + time.time() - t_g
            T['dec_calls'] = T.get('dec_calls', 0) + 1
        return out

    @torch.no_grad()
    def decompress(self, string):
        """qarv/model.py:531-557."""
        return self.decompress_batch([string])

    @torch.no_grad()
    def compress_file(self, img_path, output_path, lmb=None):
        """qarv/model.py:559-570."""
        from PIL import Image
        img = Image.open(img_path)
        img_padded = coding.pad_divisible_by(img, div=self.max_stride)
        im = coding.pil_to_tensor01(img_padded).unsqueeze_(0).to(device=self._dummy.device)
        body_str = self.compress(im, lmb=lmb)
        header_str = struct.pack('2H', img.height, img.width)
        with open(output_path, 'wb') as f:
            f.write(header_str + body_str)

    @torch.no_grad()
    def decompress_file(self, bits_path):
        """qarv/model.py:572-581."""
        with open(bits_path, 'rb') as f:
            header_str = f.read(4)
            body_str = f.read()
        img_h, img_w = struct.unpack('2H', header_str)
        im_hat = self.decompress(body_str)
        return im_hat[:, :, :img_h, :img_w]

    @torch.no_grad()
    def compress_files(self, img_paths, output_paths, lmb=None, images=None):
        """Batched compress_file: images whose PADDED sizes agree are coded by one compress_batch call (GPU work batched, the B x 9
        rANS streams coded in parallel); every output file is byte-identical to what compress_file writes for that image."""
        from PIL import Image
        imgs = images if images is not None else [Image.open(p) for p in img_paths]      # `images`: already decoded PIL images
        ims = [coding.pil_to_tensor01(coding.pad_divisible_by(img, div=self.max_stride)) for img in imgs]
        assert all(t.shape == ims[0].shape for t in ims), 'compress_files: padded sizes differ'
        bodies = self.compress_batch(torch.stack(ims).to(device=self._dummy.device), lmb=lmb)
        for img, body, out in zip(imgs, bodies, output_paths):
            with open(out, 'wb') as f:
                f.write(struct.pack('2H', img.height, img.width) + body)

    @torch.no_grad()
    def decompress_files(self, bits_paths):
        """Batched decompress_file for files of one latent shape and lambda -> list of (1,3,h,w) tensors (cropped)."""
        heads, bodies = [], []
        for p in bits_paths:
            with open(p, 'rb') as f:
                heads.append(struct.unpack('2H', f.read(4)))
                bodies.append(f.read())
        out = self.decompress_batch(bodies)
        return [out[i:i + 1, :, :h, :w] for i, (h, w) in enumerate(heads)]

    # ---- coder-free paths
    @torch.no_grad()
    @on_model_device
    def estimate(self, im, lmb=None):
        """Eval-mode forward (forward_end2end in eval mode, qarv/model.py:94-97,294-315) without entropy coding:
        returns (im_hat (B,3,H,W) in [0,1], nats (num_latents, B) float64 = sum(-ln P) per latent block and image)."""
        lmb = lmb or self.default_lmb
        B, _, H, W = im.shape
        assert (H % self.max_stride == 0) and (W % self.max_stride == 0)
        self._prepare(); self._set_lmb(lmb)
        enc = self._plan('encb', B, H, W)
        dec = self._plan('dec', B, H // self.max_stride, W // self.max_stride)
        enc.im.view(B, 3, H, W).copy_(im)
        enc.nats.zero_()
        enc.run()
        enc.fetch_status()
        torch.cuda.current_stream(enc.device).synchronize()
        enc.raise_if_flagged(where='in estimate() (encoder)')
        # the encoder stops at CompresionStopFlag; reconstruct by feeding its symbols to the decode plan (same latent layout)
        dec.sym_all.copy_(enc.sym_all)
        dec.run()
        dec.fetch_status()
        torch.cuda.current_stream(dec.device).synchronize()
        dec.raise_if_flagged(where='in estimate() (decoder)')
        return dec.out.clone(), enc.nats.view(self.num_latents, B).clone()

    @torch.no_grad()
    @on_model_device
    def conditional_sample(self, lmb, latents, emb=None, bhw_repeat=None, t=1.0, seed=None, return_latents=False):
        """Decoder output conditioned on a list of latents (qarv/model.py:365-395).  latents[i] is a (B, z_i, h_i, w_i) tensor on
        the model device (integer + prior mean, what `get_latents` / the decoder produce) or None; a missing latent is drawn from
        the prior at temperature t, z = pm + pv*N(0,1)*t + U(-.5,.5)*t (:98-100), by the device RNG of `lvae_prior_sample_f32`
        (Philox4x32-10 keyed by `seed`; default: a fresh seed per call from torch's CPU generator).  t = 0 is deterministic.
        `emb` is accepted for signature compatibility and must be None (the embedding is always derived from lmb).
        return_latents=True (not in the reference) also returns the latents actually used, [(B, z_i, h_i, w_i)]."""
        assert emb is None, 'explicit embeddings are not supported: pass lmb'
        assert len(latents) == self.num_latents
        if latents[0] is None:
            assert bhw_repeat is not None, 'bhw_repeat should be provided'
            B, nH, nW = bhw_repeat
        else:
            B, _, nH, nW = latents[0].shape
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._prepare(); self._set_lmb(float(lmb))
        pl = self._plan('dec', B, nH, nW)
        st = ctypes.c_void_p(torch.cuda.current_stream(pl.device).cuda_stream)
        lo, used = 0, []
        for li, cut in enumerate(pl.cuts):
            pl.run(lo, cut)
            zdim, hw = pl.lat_shapes[li]
            if latents[li] is None:
                # the launch at `cut` is this block's dequantize: replaced by a draw from the prior written to the same buffer
                rc = pl.lib.lvae_prior_sample_f32(pl.prm_ptrs[li], pl.zhat_ptrs[li], B * hw, zdim, zdim, float(t),
                                                  seed, li << 40, st)
                if rc:
                    raise RuntimeError(f'lvae_prior_sample_f32 failed: {rc}')
                lo = cut + 1
                if return_latents:
                    zs = pl.zhat_bufs[li][:B * hw * zdim].view(B, hw, zdim)
                    used.append(zs.permute(0, 2, 1).reshape(B, zdim, *pl.lat_hw[li]).clone())
            else:
                assert tuple(latents[li].shape) == (B, zdim, *pl.lat_hw[li]), f'latent {li}: shape {tuple(latents[li].shape)}'
                # the supplied latent is used VERBATIM (qarv/model.py:101-103, `z = latent`): it goes straight into this block's
                # z buffer (NCHW -> NHWC rows) and the dequantize launch at `cut` is skipped -- no re-quantisation against the
                # current prior mean, so edited / interpolated latents and the 'exclude' / 'reverse' / 'single' modes of
                # scripts/qarv/robust-decoding.py behave as in the reference
                zt = latents[li].to(pl.device, torch.float32).permute(0, 2, 3, 1).reshape(-1)
                pl.zhat_bufs[li][:zt.numel()].copy_(zt)
                lo = cut + 1
                if return_latents:
                    used.append(latents[li])
        pl.run(lo, None)
        pl.fetch_status()
        torch.cuda.current_stream(pl.device).synchronize()
        if all(z is not None for z in latents) or float(t) == 0.0:
            pl.raise_if_flagged(where='in conditional_sample()')
        else:
            # latents drawn from the prior at t > 0 are random numbers of the model's own scale (with untrained weights the deeper blocks'
            # prior scales are astronomically large): whatever they lead to is the sample, as in the reference -- clear the word, no error
            pl.status.zero_(); pl.status_host.zero_()
        return (pl.out.clone(), used) if return_latents else pl.out.clone()

    @torch.no_grad()
    def unconditional_sample(self, lmb, bhw_repeat, t=1.0, seed=None, return_latents=False):
        """qarv/model.py:397-404: generate images from the prior alone."""
        return self.conditional_sample(lmb, [None] * self.num_latents, bhw_repeat=bhw_repeat, t=t, seed=seed,
                                       return_latents=return_latents)

    @torch.no_grad()
    def get_latents(self, im, lmb=None):
        """What scripts/qarv/robust-decoding.py reads from forward_end2end(..., get_latent=True) in eval mode
        (qarv/model.py:94-97,294-315): per latent block the quantized latent z = symbols + prior mean as a (B, z, h, w) tensor and
        its rate in nats per image, (num_latents, B)."""
        lmb = lmb or self.default_lmb
        B, _, H, W = im.shape
        _, nats = self.estimate(im, lmb)
        dec = self._plan('dec', B, H // self.max_stride, W // self.max_stride)
        zs = []
        for li, (zdim, hw) in enumerate(dec.lat_shapes):
            o = dec.sym_off[li]
            sym = dec.sym_all[o:o + B * zdim * hw].view(B, zdim, hw).float()
            pm = dec.pm_bufs[li].view(B, hw, zdim).permute(0, 2, 1)
            zs.append((sym + pm).reshape(B, zdim, *dec.lat_hw[li]).contiguous())
        return zs, nats

    @torch.no_grad()
    def _self_evaluate(self, img_paths, lmb: float):
        """qarv/model.py:427-473 (per-image loop; estimated bpp from the likelihoods, PSNR on the cropped reconstruction)."""
        from PIL import Image
        tot = {'loss': 0.0, 'bpp': 0.0, 'psnr': 0.0}
        for impath in img_paths:
            img = Image.open(impath)
            h, w = img.height, img.width
            im = coding.pil_to_tensor01(coding.pad_divisible_by(img, div=self.max_stride)).unsqueeze_(0).to(self._dummy.device)
            im_hat, nats = self.estimate(im, lmb)
            kl = float(nats.sum()) / (3 * h * w)                                   # nats per (original) dimension
            real = coding.pil_to_tensor01(img).to(im_hat.device)
            fake = im_hat[0, :, :h, :w]
            mse = float((real - fake).square().mean())
            distortion = 4.0 * mse      # mse between (x_hat, x_target) in (-1,1) units; uses the CLAMPED reconstruction
            tot['loss'] += kl + lmb * distortion
            tot['bpp'] += kl * self.log2_e * 3
            tot['psnr'] += -10 * math.log10(mse)
        n = len(img_paths)
        out = {k: v / n for k, v in tot.items()}
        out['lambda'] = lmb
        return out

    @torch.no_grad()
    def self_evaluate(self, img_dir, lmb_range=None, steps=8, log_dir=None):
        """qarv/model.py:491-507: estimated-rate RD sweep over `steps` lambdas log-spaced in lmb_range."""
        from collections import defaultdict
        from pathlib import Path
        img_paths = sorted(Path(img_dir).rglob('*.*'))
        start, end = self.lmb_range if (lmb_range is None) else lmb_range
        lambdas = torch.linspace(math.log(start), math.log(end), steps=steps).exp().tolist()
        stats = defaultdict(list)
        for lmb in lambdas:
            for k, v in self._self_evaluate(img_paths, lmb).items():
                stats[k].append(v)
        return stats

    # ---- debugging / test access (not on the hot path)
    @torch.no_grad()
    @on_model_device
    def encode_trace(self, im, lmb=None, full=False, force_z=None):
        """Run the encode plan and return per-block int arrays (symbols, indexes in NCHW order) for parity tests.
        full=True adds the float tensors behind them per block -- pm, lv (raw log-variance parameter), qm, as (B, z, hw) arrays --
        and force_z (a list of (B, z, h, w) tensors or None per block) replaces the latent a block hands on to the blocks below
        it (teacher forcing: with the oracle's latents every block sees the oracle's inputs up to rounding noise, so a flip is
        never the cascade of an earlier one)."""
        lmb = lmb or self.default_lmb
        B, _, H, W = im.shape
        self._prepare(); self._set_lmb(lmb)
        pl = self._plan('enc', B, H, W)
        pl.im.view(B, 3, H, W).copy_(im)
        if full or force_z is not None:
            return self._trace_blocks(pl, B, force_z)
        pl.run()
        pl.fetch_status()
        torch.cuda.current_stream(pl.device).synchronize()
        pl.raise_if_flagged(where='(encode trace)')
        sym, idx = pl.sym_all.cpu().numpy(), pl.idx_all.cpu().numpy()
        out = []
        h, w = H // 64, W // 64
        for li, (z, hw) in enumerate(pl.lat_shapes):
            o = pl.sym_off[li]
            out.append(dict(symbols=sym[o:o + B * z * hw].reshape(B, z, hw).copy(),
                            indexes=idx[o:o + B * z * hw].reshape(B, z, hw).copy()))
        return out
