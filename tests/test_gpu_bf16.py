"""-m gpu: the reduced-precision GEMM mode (BASELINE.json configs[4]: bf16 MFMA for the channel-mixing GEMMs).
Kernel level: exact against an fp64 product of the bf16-ROUNDED operands (checks the MFMA lane/k mapping, not just a
loose tolerance).  Model level: encode/decode stay self-consistent and PSNR / bpp stay within a stated tolerance of the
fp32 parity path."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import seeded_init

pytestmark = pytest.mark.gpu


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('M,N,K,epi', [(300, 192, 384, 0), (1000, 384, 192, 1), (513, 128, 8, 3), (2048, 448, 256, 2),
                                       (129, 96, 1024, 0), (4096, 512, 2048, 0), (777, 48, 128, 0)])
def test_gemm_bf16_exact_vs_rounded_operands(M, N, K, epi):
    from lvae import _native
    L = _native.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W16 = Wt.to(torch.bfloat16).contiguous()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = torch.full((M, N), float('nan'), device='cuda')
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), K, K, Wt.data_ptr(), W16.data_ptr(), K
    d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec = M, N, K, epi, 1
    assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
    torch.cuda.synchronize()
    ref = A.to(torch.bfloat16).double() @ W16.double().t() + bias.double()
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = res.double() + gamma.double() * ref
    elif epi == 3:
        ref = res.double() + ref
    assert (out.double() - ref).abs().max().item() < 3e-5
    # asymmetric identity check (transposed C-write / wrong k mapping would fail it)
    if K <= 256 and epi == 0:
        Ai = torch.eye(K, device='cuda')
        o2 = torch.empty(K, N, device='cuda')
        d.A0, d.M, d.out, d.bias = Ai.data_ptr(), K, o2.data_ptr(), None
        assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        assert torch.equal(o2, W16.float().t().contiguous())


def test_model_bf16_mode(product_model):
    """qarv_base with gemm_precision='bf16': round trip still exact w.r.t. its own latents; PSNR within 0.05 dB and bpp
    within 2% of the fp32 path on the same images (random-init weights, 'wide' profile)."""
    m = product_model
    ims = torch.cat([seeded(256, 384, s) for s in (50, 51)], 0).cuda()
    lmb = 512.0
    base_mode = m._prec
    s32 = m.compress_batch(ims, lmb)
    x32 = m.decompress_batch(s32)
    try:
        m.set_gemm_precision('bf16')
        s16 = m.compress_batch(ims, lmb)
        x16 = m.decompress_batch(s16)
        xe, _ = m.estimate(ims, lmb)
        assert torch.equal(x16, xe)                       # coder + enc/dec prior consistency in the reduced-precision mode
        assert s16 == m.compress_batch(ims, lmb)
    finally:
        m.set_gemm_precision(base_mode)
    assert m.compress_batch(ims, lmb) == s32              # switching back restores the parity path bit for bit

    def psnr(x):
        return -10 * math.log10(float((x - ims).square().mean()))
    bpp32 = np.mean([len(s) for s in s32]) * 8 / (256 * 384)
    bpp16 = np.mean([len(s) for s in s16]) * 8 / (256 * 384)
    print(f'bf16 mode: PSNR {psnr(x16):.4f} vs fp32 {psnr(x32):.4f} dB; bpp {bpp16:.4f} vs {bpp32:.4f}; max|dx| {float((x16 - x32).abs().max()):.4f}')
    assert abs(psnr(x16) - psnr(x32)) < 0.05
    assert abs(bpp16 - bpp32) / bpp32 < 0.02


def seeded(h, w, seed):
    u8 = seeded_init.synthetic_image_u8(h, w, seed)
    return torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0)


@pytest.mark.parametrize('M,N,K,epi', [(300, 192, 384, 0), (1000, 384, 192, 1), (513, 128, 8, 3), (2048, 448, 256, 2),
                                       (129, 96, 1024, 0), (4096, 512, 2048, 0), (777, 48, 128, 0), (24576, 384, 192, 1)])
def test_gemm_bf16x3_is_fp32_class(M, N, K, epi):
    """prec 2: error against an fp64 reference of the UNROUNDED fp32 operands must be of the same class as the exact fp32
    MFMA path's (<= 2x + 1e-6), on data with a wide dynamic range."""
    from lvae import _native
    from lvae.models.base import split_bf16x3, pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W3 = split_bf16x3(Wt)
    assert float((W3.float().sum(0) - Wt).abs().max()) <= 2 ** -24 * float(Wt.abs().max())
    W3 = pack_bf16x3(Wt)
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    ref = A.double() @ Wt.double().t() + bias.double()
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = res.double() + gamma.double() * ref
    elif epi == 3:
        ref = res.double() + ref
    errs = []
    for prec in (0, 2):
        out = torch.full((M, N), float('nan'), device='cuda')
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), K, K, Wt.data_ptr(), W3.data_ptr(), K
        d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec = M, N, K, epi, prec
        assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        errs.append(float((out.double() - ref).abs().max()))
    print(f'M={M} N={N} K={K}: max err fp32-MFMA {errs[0]:.3e}, bf16x3 {errs[1]:.3e}')
    assert errs[1] <= 2 * errs[0] + 1e-6


@pytest.mark.parametrize('M,N,K,epi,a_gelu', [(300, 192, 384, 0, 0), (1000, 384, 192, 1, 0), (513, 128, 32, 3, 0), (2048, 448, 256, 2, 0),
                                              (129, 96, 1024, 0, 0), (4096, 512, 2048, 0, 0), (777, 72, 128, 0, 1), (24576, 1536, 384, 1, 0),
                                              (6144, 768, 3072, 2, 0), (96, 2048, 512, 0, 1), (128, 64, 64, 0, 0)])
def test_gemm_x3_pipelined_kernel_is_bit_identical(M, N, K, epi, a_gelu):
    """gemm_x3v2_kernel (software-pipelined, buffer loads) against gemm_x3_kernel (cfg = -1): same per-accumulator MFMA
    sequence, so EVERY output bit must match -- the dispatcher may pick either depending on M, and the encoder's and the
    decoder's priors have to stay identical."""
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(M + N + K + 7)
    lda = K + 8                                            # padded leading dimension
    Ab = (torch.randn(M, lda, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W3 = pack_bf16x3(Wt)
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs = []
    for cfg in (0, -1):
        out = torch.full((M, N), float('nan'), device='cuda')
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = Ab.data_ptr(), lda, K, Wt.data_ptr(), W3.data_ptr(), K
        d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec, d.a_gelu, d.cfg = M, N, K, epi, 2, a_gelu, cfg
        assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('M,N,K0,K1,epi', [(3000, 256, 256, 384, 0), (700, 384, 384, 512, 0), (260, 512, 512, 1024, 1), (5000, 128, 16, 48, 0)])
def test_gemm_x3_concat_operand_is_bit_identical(M, N, K0, K1, epi):
    """A = [A0 | A1] along K (the fused torch.cat of post_merge, qarv/model.py:66-67) on the pipelined kernel (cfg 0) against
    gemm_x3_kernel (cfg -1), and against the same product on a materialised concatenation."""
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(M + N + K0 + K1)
    K = K0 + K1
    A0 = torch.randn(M, K0 + 4, generator=g).cuda()            # padded leading dimensions
    A1 = torch.randn(M, K1 + 8, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W3 = pack_bf16x3(Wt)
    bias = torch.randn(N, generator=g).cuda()
    outs = []
    for cfg, cat in ((0, True), (-1, True), (0, False)):
        out = torch.full((M, N), float('nan'), device='cuda')
        d = _native.GemmDesc()
        if cat:
            d.A0, d.lda0, d.K0, d.A1, d.lda1, d.K1 = A0.data_ptr(), K0 + 4, K0, A1.data_ptr(), K1 + 8, K1
        else:
            Ac = torch.cat([A0[:, :K0], A1[:, :K1]], 1).contiguous()
            d.A0, d.lda0, d.K0 = Ac.data_ptr(), K, K
        d.Wt, d.Wt16, d.ldw, d.bias, d.out, d.ldo = Wt.data_ptr(), W3.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec, d.cfg = M, N, K, epi, 2, cfg
        assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('B,H,W,Cin,N,epi,a_gelu', [(2, 24, 40, 256, 8, 0, 0), (3, 9, 13, 96, 96, 1, 1), (1, 64, 96, 384, 32, 0, 0),
                                                     (5, 16, 24, 512, 96, 0, 0), (2, 7, 5, 64, 200, 3, 0)])
def test_gemm_x3_conv3_gather_is_bit_identical(B, H, W, Cin, N, epi, a_gelu):
    """3x3-tap gather (posterior heads, qres VD blocks) on the pipelined kernel (out-of-image taps = out-of-range buffer reads) against
    gemm_x3_kernel (cfg -1) and against F.conv2d in fp64."""
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(B + H + W + Cin + N)
    M, K = B * H * W, 9 * Cin
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w4 = (torch.randn(N, Cin, 3, 3, generator=g) / K ** 0.5).cuda()
    Wt = w4.permute(0, 2, 3, 1).reshape(N, K).contiguous()          # [N][(i, j, ci)]
    W3 = pack_bf16x3(Wt)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs = []
    for cfg in (0, -1):
        out = torch.full((M, N), float('nan'), device='cuda')
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = x.data_ptr(), Cin, Cin, Wt.data_ptr(), W3.data_ptr(), K
        d.bias, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec, d.a_mode, d.H, d.W, d.a_gelu, d.cfg = M, N, K, epi, 2, 2, H, W, a_gelu, cfg
        assert L.lvae_gemm_f32(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])
    xin = F.gelu(x.double()) if a_gelu else x.double()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w4.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(M, N)
    ref = {0: ref, 1: F.gelu(ref), 3: res.double() + ref}[epi]
    assert float((outs[0].double() - ref).abs().max()) < 3e-5
