"""-m gpu: the f16x2 split-MFMA GEMM (prec 4, csrc/gemm_h2.hip): fp32-class accuracy against fp64 on wide-range data, exact
agreement with an fp64 emulation of its own arithmetic up to accumulation rounding, invariance of every output bit under batch size,
tile width, split-K and the fused gathers (what keeps encoder and decoder priors in lock-step), and the model-level golden cases."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gemm(A, lda, K0, Wt, W16, bias, out, N, M, epi=0, prec=4, gamma=None, res=None, a_gelu=0, ksplit=0, ws=None, cnt=None, K=None, **kw):
    """K0 = length of the (first) A source; K = total reduction length (default K0), also the weights' row stride."""
    from lvae import _native
    K = K or K0
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), lda, K0, Wt.data_ptr(), W16.data_ptr(), K
    d.bias, d.out, d.ldo = bias.data_ptr() if bias is not None else None, out.data_ptr(), N
    if gamma is not None:
        d.gamma = gamma.data_ptr()
    if res is not None:
        d.res, d.ldres = res.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec, d.a_gelu = M, N, K, epi, prec, a_gelu
    if ksplit > 1:
        d.ksplit, d.ws = ksplit, ws.data_ptr()
        if cnt is not None:
            d.cnt = cnt.data_ptr()
    for k, v in kw.items():
        setattr(d, k, v)
    rc = _native.lib().lvae_gemm_f32(ctypes.byref(d), _st())
    torch.cuda.synchronize()
    return rc


@pytest.mark.parametrize('scale', [1.0, 1e-4, 300.0])
@pytest.mark.parametrize('M,N,K,epi', [(300, 192, 384, 0), (1000, 384, 192, 1), (513, 128, 32, 3), (2048, 448, 256, 2),
                                       (129, 96, 1024, 0), (4096, 512, 2048, 0), (777, 48, 128, 0), (24576, 384, 192, 1)])
def test_gemm_f16x2_is_fp32_class(M, N, K, epi, scale):
    """Error against an fp64 reference of the UNROUNDED fp32 operands: of the class of the exact fp32 MFMA path's (<= 2x + 1e-6 of
    the output scale), on rows spanning e^+-3 in magnitude, for ordinary, tiny (fp16-subnormal hi terms) and large operands."""
    from lvae.models.base import pack_bf16x3, pack_f16x2, split_f16x2
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g)) * scale).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W2 = split_f16x2(Wt)
    assert float((W2[0].double() + W2[1].double() / 2048 - Wt.double()).abs().max()) <= 2 ** -23 * float(Wt.abs().max())
    Wh, W3 = pack_f16x2(Wt), pack_bf16x3(Wt)
    bias, gamma = (torch.randn(N, generator=g) * scale).cuda(), torch.rand(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * scale).cuda()
    ref = A.double() @ Wt.double().t() + bias.double()
    ref = {0: ref, 1: F.gelu(ref), 2: res.double() + gamma.double() * ref, 3: res.double() + ref}[epi]
    errs = {}
    for prec, W16 in ((0, W3), (2, W3), (4, Wh)):
        out = torch.full((M, N), float('nan'), device='cuda')
        assert _gemm(A, K, K, Wt, W16, bias, out, N, M, epi, prec, gamma=gamma, res=res) == 0
        errs[prec] = float((out.double() - ref).abs().max())
    print(f'M={M} N={N} K={K} x{scale:g}: max err fp32-MFMA {errs[0]:.3e}, bf16x3 {errs[2]:.3e}, f16x2 {errs[4]:.3e}')
    assert errs[4] <= 2 * errs[0] + 1e-6 * scale


@pytest.mark.parametrize('M,N,K', [(257, 128, 256), (1024, 64, 96 * 32), (5000, 200, 64)])
def test_gemm_f16x2_matches_its_own_arithmetic(M, N, K):
    """The kernel against an fp64 evaluation of exactly its three cross terms (H + X / 2048 of the split operands): what is left is
    the fp32 rounding of the two accumulators (<= K/16 + 2 roundings of 2^-24 relative each) -- a wrong lane / k mapping, a swapped
    plane or a missing term is orders of magnitude above it."""
    from lvae.models.base import pack_f16x2, split_f16x2
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    A2, W2 = split_f16x2(A).double(), split_f16x2(Wt).double()
    H = A2[0] @ W2[0].t()
    X = A2[1] @ W2[0].t() + A2[0] @ W2[1].t()
    ref = H + X / 2048.0
    out = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, pack_f16x2(Wt), None, out, N, M) == 0
    bound = (K / 16 + 4) * 2.0 ** -24 * (A.double().abs() @ Wt.double().abs().t())
    assert bool(((out.double() - ref).abs() <= bound).all())
    # identity: every weight comes back exactly where hi + lo' / 2048 represents it exactly (here: weights rounded to 20 bits)
    if K <= 256:
        Wq = (Wt * 1024).round() / 1024
        o2 = torch.empty(K, N, device='cuda')
        assert _gemm(torch.eye(K, device='cuda'), K, K, Wq, pack_f16x2(Wq), None, o2, N, K) == 0
        assert torch.equal(o2, Wq.t().contiguous())


def test_gemm_f16x2_bits_do_not_depend_on_launch_geometry():
    """Rows of a large problem == the same rows computed alone (batch invariance, tile-row position), columns under a 128-wide and a
    64-wide tile (N = 128 -> TN 2, N = 64 -> TN 1), a padded leading dimension, and split-K through the reduce kernel == through the
    in-kernel last-arriver reduction."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(5)
    M, N, K = 3000, 128, 768
    Ab = torch.randn(M, K + 8, generator=g).cuda()
    A = Ab[:, :K].contiguous()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Wh = pack_f16x2(Wt)
    full = torch.empty(M, N, device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, full, N, M, 1) == 0
    pad = torch.empty(M, N, device='cuda')
    assert _gemm(Ab, K + 8, K, Wt, Wh, bias, pad, N, M, 1) == 0
    assert torch.equal(full, pad)
    for r0, n in ((0, 1), (130, 200), (2999, 1), (1024, 384)):
        part = torch.empty(n, N, device='cuda')
        assert _gemm(A[r0:r0 + n].contiguous(), K, K, Wt, Wh, bias, part, N, n, 1) == 0
        assert torch.equal(part, full[r0:r0 + n])
    W64 = Wt[:64].contiguous()
    half = torch.empty(M, 64, device='cuda')
    assert _gemm(A, K, K, W64, pack_f16x2(W64), bias[:64].contiguous(), half, 64, M, 1) == 0
    assert torch.equal(half, full[:, :64].contiguous())
    # split-K: slices summed in slice order by the reduce kernel / by the tile's last arriver
    S = 4
    ws = torch.empty(S * M * N, device='cuda')
    a = torch.empty(M, N, device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, a, N, M, 1, ksplit=S, ws=ws) == 0
    cnt = torch.zeros(4096, dtype=torch.int32, device='cuda')
    b = torch.empty(M, N, device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, b, N, M, 1, ksplit=S, ws=ws, cnt=cnt) == 0
    assert torch.equal(a, b) and int(cnt.abs().sum()) == 0
    ref = F.gelu(A.double() @ Wt.double().t() + bias.double())
    assert float((a.double() - ref).abs().max()) < 3e-5


@pytest.mark.parametrize('M,N,K0,K1', [(3000, 256, 256, 384), (260, 512, 512, 1024), (5000, 128, 16, 48)])
def test_gemm_f16x2_concat_operand(M, N, K0, K1):
    """A = [A0 | A1] along K (post_merge's fused torch.cat, qarv/model.py:66-67) == the same product on a materialised concatenation."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(M + N + K0 + K1)
    K = K0 + K1
    A0, A1 = torch.randn(M, K0 + 4, generator=g).cuda(), torch.randn(M, K1 + 8, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Wh, bias = pack_f16x2(Wt), torch.randn(N, generator=g).cuda()
    o1, o2 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    assert _gemm(A0, K0 + 4, K0, Wt, Wh, bias, o1, N, M, A1=A1.data_ptr(), lda1=K1 + 8, K1=K1, K=K) == 0
    Ac = torch.cat([A0[:, :K0], A1[:, :K1]], 1).contiguous()
    assert _gemm(Ac, K, K, Wt, Wh, bias, o2, N, M) == 0
    assert torch.equal(o1, o2)
    assert float((o1.double() - (Ac.double() @ Wt.double().t() + bias.double())).abs().max()) < 3e-5


@pytest.mark.parametrize('B,H,W,Cin,N,epi,a_gelu', [(2, 24, 40, 256, 8, 0, 0), (3, 9, 13, 96, 96, 1, 1), (1, 64, 96, 384, 32, 0, 0),
                                                     (5, 16, 24, 512, 96, 0, 0), (2, 7, 5, 64, 200, 3, 0)])
def test_gemm_f16x2_conv3_gather(B, H, W, Cin, N, epi, a_gelu):
    """3x3-tap gather (posterior heads, qres VD blocks; out-of-image taps = out-of-range buffer reads) against F.conv2d in fp64, and
    image b of the batch == the same image alone."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(B + H + W + Cin + N)
    M, K = B * H * W, 9 * Cin
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w4 = (torch.randn(N, Cin, 3, 3, generator=g) / K ** 0.5).cuda()
    Wt = w4.permute(0, 2, 3, 1).reshape(N, K).contiguous()
    Wh, bias = pack_f16x2(Wt), torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(x, Cin, Cin, Wt, Wh, bias, out, N, M, epi, res=res, a_gelu=a_gelu, a_mode=2, H=H, W=W, K=K) == 0
    xin = F.gelu(x.double()) if a_gelu else x.double()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w4.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(M, N)
    ref = {0: ref, 1: F.gelu(ref), 3: res.double() + ref}[epi]
    assert float((out.double() - ref).abs().max()) < 3e-5
    one = torch.empty(H * W, N, device='cuda')
    b = B - 1
    assert _gemm(x[b].contiguous(), Cin, Cin, Wt, Wh, bias, one, N, H * W, epi, res=res[b * H * W:].contiguous(), a_gelu=a_gelu,
                 a_mode=2, H=H, W=W, K=K) == 0
    assert torch.equal(one, out[b * H * W:])


@pytest.mark.parametrize('B,H,W,Cin,N,epi,S', [(1, 50, 100, 48, 48, 1, 1), (2, 64, 96, 96, 96, 1, 1), (3, 37, 53, 48, 24, 0, 1), (2, 24, 40, 256, 8, 0, 3),
                                               (1, 128, 192, 48, 48, 1, 1), (4, 64, 96, 384, 96, 0, 6), (8, 128, 192, 48, 48, 1, 1)])
def test_gemm_h2n_equals_gemm_h2_conv3(B, H, W, Cin, N, epi, S):
    """The narrow-output kernel (csrc/gemm_h2n.hip: A fragments straight from global memory, weights streamed through LDS in chunks;
    cfg = 3 takes it wherever it applies) against gemm_h2_kernel (cfg = 1) on the 3x3-tap gather: every output bit equal -- ragged
    last workgroup, N = 24 / 48 (partial column blocks) / 96, K = 432 (27 k16 steps: a short last weight chunk) ... 3456, and the
    split-K launches (S = 3, 6: serial slices here, workspace + reduction there)."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(B + H + W + Cin + N)
    M, K = B * H * W, 9 * Cin
    x = (torch.randn(B, H, W, Cin, generator=g) * torch.exp(torch.randn(B, H, W, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Wh, bias = pack_f16x2(Wt), torch.randn(N, generator=g).cuda()
    ws = torch.empty(S * M * N, device='cuda') if S > 1 else None
    ref = torch.full((M, N), float('nan'), device='cuda')
    if K % 32 == 0:
        assert _gemm(x, Cin, Cin, Wt, Wh, bias, ref, N, M, epi, a_mode=2, H=H, W=W, K=K, ksplit=S, ws=ws, cfg=1) == 0
    else:
        # K = 16 (mod 32): gemm_h2_kernel walks the k16 steps in pairs and does not take it; the narrow kernel is the only f16x2 form
        # of such a layer (every batch size), so the reference is its own first launch + fp64 below
        assert _gemm(x, Cin, Cin, Wt, Wh, bias, ref, N, M, epi, a_mode=2, H=H, W=W, K=K, cfg=1) == -22
        assert _gemm(x, Cin, Cin, Wt, Wh, bias, ref, N, M, epi, a_mode=2, H=H, W=W, K=K) == 0
        w4 = Wt.view(N, 3, 3, Cin).permute(0, 3, 1, 2).double()
        r64 = F.conv2d(x.double().permute(0, 3, 1, 2), w4, bias.double(), padding=1).permute(0, 2, 3, 1).reshape(M, N)
        r64 = F.gelu(r64) if epi == 1 else r64
        assert float((ref.double() - r64).abs().max()) < 3e-5 * max(1.0, float(r64.abs().max()))
        one = torch.empty(H * W, N, device='cuda')                    # image b of the batch == the same image alone
        assert _gemm(x[B - 1].contiguous(), Cin, Cin, Wt, Wh, bias, one, N, H * W, epi, a_mode=2, H=H, W=W, K=K) == 0
        assert torch.equal(one, ref[(B - 1) * H * W:])
    for rep in range(2):
        out = torch.full((M, N), float('nan'), device='cuda')
        assert _gemm(x, Cin, Cin, Wt, Wh, bias, out, N, M, epi, a_mode=2, H=H, W=W, K=K, ksplit=S, ws=ws, cfg=3) == 0
        assert not torch.isnan(ref).any() and torch.equal(out.view(torch.int32), ref.view(torch.int32)), \
            f'rep {rep}: {int((out.view(torch.int32) != ref.view(torch.int32)).sum())} of {out.numel()} words differ'


@pytest.mark.parametrize('M,N,K,epi,S', [(5000, 48, 192, 1, 1), (196608, 48, 384, 1, 1), (70001, 96, 768, 1, 1), (300, 16, 64, 0, 1),
                                         (49152, 96, 384, 3, 1), (40000, 64, 1024, 2, 4), (33000, 8, 2304, 0, 2)])
def test_gemm_h2n_equals_gemm_h2_plain(M, N, K, epi, S):
    """Same for plain fp32 rows (the 1x1 reductions of the qres bottleneck blocks), every epilogue, with and without split-K."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    A[M // 3] = 0.0
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Wh, bias = pack_f16x2(Wt), torch.randn(N, generator=g).cuda()
    gamma, res = torch.rand(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    ws = torch.empty(S * M * N, device='cuda') if S > 1 else None
    ref = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, ref, N, M, epi, gamma=gamma, res=res, ksplit=S, ws=ws, cfg=1) == 0
    out = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, out, N, M, epi, gamma=gamma, res=res, ksplit=S, ws=ws, cfg=3) == 0
    assert not torch.isnan(ref).any() and torch.equal(out.view(torch.int32), ref.view(torch.int32)), \
        f'{int((out.view(torch.int32) != ref.view(torch.int32)).sum())} of {out.numel()} words differ'
    # the shape rule (cfg = 0) may pick either kernel: same bits
    out2 = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, Wh, bias, out2, N, M, epi, gamma=gamma, res=res, ksplit=S, ws=ws) == 0
    assert torch.equal(out2.view(torch.int32), ref.view(torch.int32))


def test_gemm_f16x2_rejects_what_it_does_not_take():
    from lvae.models.base import pack_f16x2
    A, Wt = torch.randn(64, 48).cuda(), torch.randn(32, 48).cuda()
    assert pack_f16x2(torch.full((4, 32), 1e5)) is None               # beyond fp16's range: the host keeps such a GEMM on bf16x3
    out = torch.empty(64, 32, device='cuda')
    assert _gemm(A, 48, 48, Wt, pack_f16x2(Wt), None, out, 32, 64) == 0        # K = 16 (mod 32), N <= 96: the narrow-output kernel (gemm_h2n.hip)
    assert float((out.double() - A.double() @ Wt.double().t()).abs().max()) < 1e-5
    W2 = torch.randn(200, 48).cuda()
    out2 = torch.empty(64, 200, device='cuda')
    assert _gemm(A, 48, 48, W2, pack_f16x2(W2), None, out2, 200, 64) == -22    # K = 16 (mod 32) with a wide output: no f16x2 kernel
    A3, W3 = torch.randn(64, 40).cuda(), torch.randn(32, 40).cuda()
    assert pack_f16x2(W3) is None                                              # K % 16 != 0: no f16x2 weight format


# ---------------------------------------------------------------------------------------------- pre-split operands (csrc/gemm_h2p.hip)
@pytest.mark.parametrize('tile', [42, 41, 22, 21, 23, 0])
@pytest.mark.parametrize('M,N,K,epi', [(1000, 384, 192, 1), (256, 128, 32, 0), (3001, 192, 384, 2), (520, 448, 256, 1), (12288, 768, 384, 1),
                                       (777, 64, 1536, 2), (40000, 384, 128, 1), (70001, 192, 384, 2)])
def test_gemm_h2p_equals_h2_bit_for_bit(M, N, K, epi, tile):
    """Both operands pre-split (H2K32 planes, LDS-DMA main loop) against gemm_h2_kernel on the fp32 operand: same split, same
    per-accumulator MFMA sequence => every output bit equal, for every tile shape (ragged M / N included)."""
    from lvae.models.base import pack_f16x2, pack_f16x2_k32
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    ref = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, pack_f16x2(Wt), bias, ref, N, M, epi, gamma=gamma, res=res) == 0
    out = torch.full((M, N), float('nan'), device='cuda')
    Ah = pack_f16x2_k32(A)
    assert _gemm(Ah, K, K, Wt, pack_f16x2_k32(Wt), bias, out, N, M, epi, gamma=gamma, res=res, a_h2=1, cfg=tile) == 0
    assert not torch.isnan(ref).any() and torch.equal(out, ref)


@pytest.mark.parametrize('a_h2', [0, 1])
@pytest.mark.parametrize('M,N,K,epi', [(1000, 384, 192, 1), (300, 64, 64, 0), (4100, 768, 384, 1)])
def test_gemm_out_h2_is_the_split_of_the_fp32_result(M, N, K, epi, a_h2):
    """out_h2: the stored planes are exactly split_f16x2 of what the same GEMM stores as fp32 (fc1's hidden map for fc2)."""
    from lvae.models.base import pack_f16x2, pack_f16x2_k32, split_f16x2, unpack_f16x2_k32
    g = torch.Generator().manual_seed(M + N + K + 3)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Ain, Wp = (pack_f16x2_k32(A), pack_f16x2_k32(Wt)) if a_h2 else (A, pack_f16x2(Wt))
    ref = torch.empty(M, N, device='cuda')
    assert _gemm(Ain, K, K, Wt, Wp, bias, ref, N, M, epi, a_h2=a_h2) == 0
    out = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(Ain, K, K, Wt, Wp, bias, out, N, M, epi, a_h2=a_h2, out_h2=1) == 0
    hi, lo = unpack_f16x2_k32(out, M, N)
    want = split_f16x2(ref)
    assert torch.equal(hi, want[0]) and torch.equal(lo, want[1])


@pytest.mark.parametrize('B,H,W,C,k,affine', [(2, 16, 24, 192, 7, 'adaln'), (1, 9, 13, 128, 7, 'adaln'), (3, 8, 8, 512, 3, 'adaln'),
                                               (2, 12, 20, 384, 5, 'ln'), (1, 4, 6, 512, 1, 'adaln'), (2, 33, 17, 256, 7, 'none')])
def test_dwconv_ln_h2_is_the_split_of_the_fp32_result(B, H, W, C, k, affine):
    """lvae_dwconv_ln_h2 (depthwise + LayerNorm + affine, result stored pre-split for fc1) == split_f16x2 of lvae_dwconv_ln_f32."""
    from lvae import _native
    from lvae.models.base import split_f16x2, unpack_f16x2_k32
    L = _native.lib()
    g = torch.Generator().manual_seed(B + H + W + C + k)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    wt = (torch.randn(k * k, C, generator=g) / k).cuda()
    bias = torch.randn(C, generator=g).cuda()
    a0, a1 = torch.randn(C, generator=g).cuda(), (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    ln = (a1.data_ptr(), a0.data_ptr()) if affine == 'ln' else (None, None)
    ada = (a0.data_ptr(), a1.data_ptr()) if affine == 'adaln' else (None, None)
    y = torch.empty_like(x)
    assert L.lvae_dwconv_ln_f32(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), *ln, *ada, y.data_ptr(), B, H, W, C, k, _st()) == 0
    y2 = torch.full_like(x, float('nan'))
    assert L.lvae_dwconv_ln_h2(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), *ln, *ada, y2.data_ptr(), B, H, W, C, k, _st()) == 0
    torch.cuda.synchronize()
    hi, lo = unpack_f16x2_k32(y2, B * H * W, C)
    want = split_f16x2(y.view(-1, C))
    assert torch.equal(hi, want[0]) and torch.equal(lo, want[1])
    assert L.lvae_dwconv_ln_h2(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), None, None, None, None, y2.data_ptr(), B, H, W, 144, k, _st()) == -22


@pytest.mark.parametrize('B,Ho,Wo,Cin,Cout', [(2, 6, 10, 192, 384), (1, 3, 5, 512, 512), (3, 1, 1, 384, 512), (2, 16, 24, 384, 512), (1, 7, 9, 8, 64)])
def test_gemm_f16x2_patch2_gather(B, Ho, Wo, Cin, Cout):
    """2x2 / stride-2 patch gather (patch_downsample, common.py:29-30) on the f16x2 kernel against F.conv2d in fp64, and image b of the
    batch == the same image alone."""
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = torch.randn(B, Cin, 2 * Ho, 2 * Wo, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 2, 2, generator=g) / (4 * Cin) ** 0.5).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=2).permute(0, 2, 3, 1).reshape(-1, Cout)
    xn = x.permute(0, 2, 3, 1).contiguous()
    Wt = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    M, K = B * Ho * Wo, 4 * Cin
    out = torch.full((M, Cout), float('nan'), device='cuda')
    assert _gemm(xn, Cin, Cin, Wt, pack_f16x2(Wt), bias, out, Cout, M, a_mode=1, H=Ho, W=Wo, K=K) == 0
    assert float((out.double() - ref).abs().max()) < 2e-5
    one = torch.empty(Ho * Wo, Cout, device='cuda')
    assert _gemm(xn[B - 1].contiguous(), Cin, Cin, Wt, pack_f16x2(Wt), bias, one, Cout, Ho * Wo, a_mode=1, H=Ho, W=Wo, K=K) == 0
    assert torch.equal(one, out[(B - 1) * Ho * Wo:])


@pytest.mark.parametrize('M,N,K,S,epi', [(3072, 1024, 512, 4, 1), (3072, 512, 1024, 8, 2), (1536, 1536, 512, 4, 1), (1536, 512, 1536, 6, 2),
                                         (777, 64, 2048, 16, 0), (2000, 192, 256, 2, 3), (384, 512, 1024, 8, 2), (5000, 96, 128, 2, 1)])
def test_gemm_h2p_serial_split_k_equals_parallel_split_k(M, N, K, S, epi):
    """Serial split-K (gemm_h2p FOLD: one workgroup walks the S slices and adds their partial sums in slice order, pre-split operands)
    against the parallel form on the fp32 operand (gemm_h2_kernel with S slice workgroups per tile + the reduce kernel, and + the
    in-kernel last arriver): the same operations in the same order => every output bit equal, also with exact zeros and ragged M / N.
    This is what lets the host pick the form by BATCH size (tiles available) while the slice count stays a per-image rule."""
    from lvae.models.base import pack_f16x2, pack_f16x2_k32, unpack_f16x2_k32
    g = torch.Generator().manual_seed(M + N + K + S)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    A[5] = 0.0                                                       # a row of exact zeros (signed-zero handling of the partial sums)
    A[:, 7] = 0.0
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    bias[3] = 0.0
    res = torch.randn(M, N, generator=g).cuda()
    ws = torch.empty(S * M * N, device='cuda')
    ref = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, pack_f16x2(Wt), bias, ref, N, M, epi, gamma=gamma, res=res, ksplit=S, ws=ws) == 0
    cnt = torch.zeros(((M + 63) // 64) * ((N + 31) // 32), dtype=torch.int32, device='cuda')
    ref2 = torch.full((M, N), float('nan'), device='cuda')
    assert _gemm(A, K, K, Wt, pack_f16x2(Wt), bias, ref2, N, M, epi, gamma=gamma, res=res, ksplit=S, ws=ws, cnt=cnt) == 0
    assert not torch.isnan(ref).any() and torch.equal(ref, ref2)
    Ah, Wh = pack_f16x2_k32(A), pack_f16x2_k32(Wt)
    out = torch.full((M, N), float('nan'), device='cuda')
    ws.fill_(float('nan'))                                           # the serial form uses no workspace
    assert _gemm(Ah, K, K, Wt, Wh, bias, out, N, M, epi, gamma=gamma, res=res, a_h2=1, ksplit=S, ws=ws) == 0
    assert not torch.isnan(out).any() and torch.equal(out, ref)
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32)) or int((out.view(torch.int32) != ref.view(torch.int32)).sum()) == \
        int(((out == 0) & (ref == 0) & (out.view(torch.int32) != ref.view(torch.int32))).sum())       # at most signs of exact zeros
    if N % 32 == 0 and epi in (0, 1):
        planes = torch.empty(M, N, device='cuda')                    # H2K32 planes: 4 bytes per element
        assert _gemm(Ah, K, K, Wt, Wh, bias, planes, N, M, epi, a_h2=1, out_h2=1, ksplit=S, ws=ws) == 0
        got, want = unpack_f16x2_k32(planes, M, N), unpack_f16x2_k32(pack_f16x2_k32(ref), M, N)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


@pytest.mark.parametrize('C,HID', [(192, 384), (384, 768)])
@pytest.mark.parametrize('M', [128, 1000, 300 * 128 + 5, 98304, 196608])
def test_mlp_h2c_equals_two_gemms(M, C, HID):
    """The hidden-chunked fused MLP of the C = 192 / hidden = 384 blocks and -- 64-row tiles, the W2 rows of a k32 step in three parts --
    of the C = 384 / hidden = 768 blocks (csrc/mlp_h2c.hip: a persistent workgroup per CU walks the hidden
    dimension in chunks of 128 -- fc1 stages, GELU / split into LDS, fc2 stages -- with every operand streamed by LDS-DMA through one flat
    ring) against the two pre-split GEMM launches it replaces: every output bit equal.  M = 128: one tile; 1000 / 38405: ragged last
    tile, fewer tiles than CUs / more than one tile per workgroup; 98304 / 196608: the model's launches (384 / 768 tiles: several tiles
    per persistent workgroup, the ring running across tile boundaries).  In place (out aliasing the residual) like the plans use it."""
    from lvae import _native
    from lvae.models.base import pack_f16x2_k32
    if C == 384 and M == 196608:
        M = 49152 + 64 * 7 + 3                                         # the stride-8 map of the model, plus a ragged tail of 64-row tiles
    g = torch.Generator().manual_seed(M)
    yf = (torch.randn(M, C, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    yf[3] = 0.0
    W1 = (torch.randn(HID, C, generator=g) / C ** 0.5).cuda()
    W2 = (torch.randn(C, HID, generator=g) / HID ** 0.5).cuda()
    b1, b2, gamma = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda()
    res = torch.randn(M, C, generator=g).cuda()
    y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
    hid = torch.empty(M, HID, device='cuda')                           # H2K32 planes, 4 bytes per element
    ref = torch.full((M, C), float('nan'), device='cuda')
    assert _gemm(y, C, C, W1, w1h, b1, hid, HID, M, 1, a_h2=1, out_h2=1) == 0
    assert _gemm(hid, HID, HID, W2, w2h, b2, ref, C, M, 2, gamma=gamma, res=res, a_h2=1) == 0
    d = _native.MlpDesc()
    out = res.clone()                                                # in place: out aliases the residual
    d.y, d.w1, d.b1, d.w2, d.b2, d.gamma = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr()
    d.res, d.out, d.M, d.C, d.hid = out.data_ptr(), out.data_ptr(), M, C, HID
    for rep in range(3):                                              # repeated launches: nothing may depend on what the last one left in LDS
        out.copy_(res)
        assert _native.lib().lvae_mlp_h2f(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        assert not torch.isnan(ref).any() and torch.equal(out, ref), f'rep {rep}: {int((out != ref).sum())} of {out.numel()} elements differ'
    ref64 = res.double() + gamma.double() * (F.gelu(yf.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double())
    assert float((out.double() - ref64).abs().max()) < 1e-4 * float(ref64.abs().max())


@pytest.mark.parametrize('C,HID,S1,S2', [(512, 2048, 4, 16), (512, 1024, 4, 8), (512, 1536, 2, 8), (512, 2048, 2, 8), (512, 1024, 1, 8)])
@pytest.mark.parametrize('M', [96, 100, 384, 1536, 3072 + 5])
def test_mlp_sk_equals_split_k_gemms(M, C, HID, S1, S2):
    """The small-map MLP (csrc/mlp_sk.hip: workgroup (32 rows, fc2 slice c) computes its hid / S2 hidden columns itself -- fc1's S1 slices
    folded in order -- and writes fc2's partial sums to plane c; then the split-K reduce launch) against the two serial split-K launches
    it replaces (gemm_h2p FOLD: fc1 with ksplit = S1 and the pre-split GELU epilogue, fc2 with ksplit = S2, gamma + residual), which
    tests above tie to the parallel split-K form: every output bit equal.  Shapes: the model's stride-64 / 32 blocks (hidden 2048 / 16
    slices, 1024 / 8, 1536 / 8 with its 192-column chunks and a partial third output group), a 1216x1216 image's stride-64 block (2048 /
    8: 256-column chunks), S1 = 1; M = 96 / 384 / 1536: one and four images, 100 / 3077: ragged last row tile.  In place like the plans."""
    from lvae import _native
    from lvae.models.base import pack_f16x2_k32
    assert _native.lib().lvae_mlp_sk_supported(C, HID, S1, S2) == 1
    g = torch.Generator().manual_seed(M + HID + S1)
    yf = (torch.randn(M, C, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    yf[3] = 0.0
    W1 = (torch.randn(HID, C, generator=g) / C ** 0.5).cuda()
    W2 = (torch.randn(C, HID, generator=g) / HID ** 0.5).cuda()
    b1, b2, gamma = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda()
    b1[5] = 0.0
    res = torch.randn(M, C, generator=g).cuda()
    y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
    hid = torch.empty(M, HID, device='cuda')                           # H2K32 planes, 4 bytes per element
    ref = torch.full((M, C), float('nan'), device='cuda')
    ws = torch.full((S2 * M * C,), float('nan'), device='cuda')
    assert _gemm(y, C, C, W1, w1h, b1, hid, HID, M, 1, a_h2=1, out_h2=1, ksplit=S1, ws=ws) == 0
    assert _gemm(hid, HID, HID, W2, w2h, b2, ref, C, M, 2, gamma=gamma, res=res, a_h2=1, ksplit=S2, ws=ws) == 0
    d = _native.MlpSkDesc()
    out = res.clone()
    d.y, d.w1, d.b1, d.w2, d.b2, d.gamma = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr()
    d.res, d.out, d.ws, d.M, d.C, d.hid, d.S1, d.S2 = out.data_ptr(), out.data_ptr(), ws.data_ptr(), M, C, HID, S1, S2
    for rep in range(3):
        out.copy_(res)
        ws.fill_(float('nan'))
        assert _native.lib().lvae_mlp_sk(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        assert not torch.isnan(ref).any() and torch.equal(out, ref), f'rep {rep}: {int((out != ref).sum())} of {out.numel()} elements differ'
    ref64 = res.double() + gamma.double() * (F.gelu(yf.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double())
    assert float((out.double() - ref64).abs().max()) < 1e-4 * float(ref64.abs().max())


@pytest.mark.parametrize('M', [5, 128, 129, 1000, 24576 + 77, 256 * 128 + 1, 98304, 196608])
def test_mlp_h2f_equals_two_gemms(M):
    """The fused MLP of the C = 128 / hidden = 192 blocks (lvae_mlp_h2f -> csrc/mlp_h2c.hip <128, 192, 64>: fc1 -> GELU -> fc2 in one launch, hidden chunks of 64 in LDS)
    against the two pre-split GEMM launches it replaces (fc1 with the pre-split GELU epilogue, fc2 with gamma + residual): every
    output bit equal -- ragged M, wide-range rows, zero rows -- and in place (out aliasing the residual) like the plans use it."""
    from lvae import _native
    from lvae.models.base import pack_f16x2_k32
    C, HID = 128, 192
    g = torch.Generator().manual_seed(M)
    yf = (torch.randn(M, C, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    yf[3] = 0.0
    W1 = (torch.randn(HID, C, generator=g) / C ** 0.5).cuda()
    W2 = (torch.randn(C, HID, generator=g) / HID ** 0.5).cuda()
    b1, b2, gamma = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda()
    res = torch.randn(M, C, generator=g).cuda()
    y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
    hid = torch.empty(M, HID, device='cuda')                           # H2K32 planes, 4 bytes per element
    ref = torch.full((M, C), float('nan'), device='cuda')
    assert _gemm(y, C, C, W1, w1h, b1, hid, HID, M, 1, a_h2=1, out_h2=1) == 0
    assert _gemm(hid, HID, HID, W2, w2h, b2, ref, C, M, 2, gamma=gamma, res=res, a_h2=1) == 0
    d = _native.MlpDesc()
    out = res.clone()                                                # in place: out aliases the residual
    d.y, d.w1, d.b1, d.w2, d.b2, d.gamma = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr()
    d.res, d.out, d.M, d.C, d.hid = out.data_ptr(), out.data_ptr(), M, C, HID
    for rep in range(3):                                              # repeated launches: nothing may depend on what the last one left in LDS
        out.copy_(res)                                               # (the tile's A rows are resident in LDS since round 5, fetched one tile ahead)
        assert _native.lib().lvae_mlp_h2f(ctypes.byref(d), _st()) == 0
        torch.cuda.synchronize()
        assert not torch.isnan(ref).any() and torch.equal(out, ref), f'rep {rep}: {int((out != ref).sum())} of {out.numel()} elements differ'
    ref64 = res.double() + gamma.double() * (F.gelu(yf.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double())
    assert float((out.double() - ref64).abs().max()) < 1e-4 * float(ref64.abs().max())
    d.hid = 256
    assert _native.lib().lvae_mlp_h2f(ctypes.byref(d), _st()) == -22

