"""-m gpu: each HIP kernel (called through the C ABI via ctypes) against a plain PyTorch fp32 reference of the same op.

Tolerances: fp32 MFMA accumulation is an fmaf chain in a fixed k-order; torch sums in another order, so outputs agree
to ~1e-6 relative of sum|a*b| -- asserted as 2e-5 abs on O(1) data (the north-star tolerance is 1e-4 end to end).
"""
import ctypes

import numpy as np
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    from lvae import _native
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _native.lib()


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gemm(L, **kw):
    from lvae._native import GemmDesc
    d = GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    rc = L.lvae_gemm_f32(ctypes.byref(d), _st())
    assert rc == 0, rc
    torch.cuda.synchronize()


@pytest.mark.parametrize('M,N,K', [(96, 2048, 512), (1000, 512, 2048), (24576, 384, 192), (777, 48, 128), (300, 64, 512),
                                   (513, 96, 384), (129, 16, 256), (2050, 192, 128), (64, 448, 256), (4096, 256, 8)])
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
def test_gemm_plain(L, M, N, K, epi):
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N * 3 + K + epi)
    A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    out = torch.full((M, N), float('nan'), device='cuda')
    _gemm(L, A0=A, lda0=K, K0=K, Wt=Wt, ldw=K, bias=bias, gamma=gamma, res=res, ldres=N, out=out, ldo=N, M=M, N=N, K=K,
          a_mode=0, epi=epi, store=0)
    ref = A.double() @ Wt.double().t() + bias.double()
    if epi == 1:
        ref = F.gelu(ref)
    elif epi == 2:
        ref = res.double() + gamma.double() * ref
    elif epi == 3:
        ref = res.double() + ref
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5, err


def test_gemm_asymmetric_identity(L):
    """A = I with an asymmetric W catches a transposed C-write (cdna guide, 'always A=I-check with asymmetric B')."""
    K = N = 128
    A = torch.eye(K, device='cuda')
    Wt = (torch.arange(N * K, device='cuda', dtype=torch.float32).reshape(N, K) % 251) / 251
    out = torch.empty(K, N, device='cuda')
    _gemm(L, A0=A, lda0=K, K0=K, Wt=Wt, ldw=K, bias=torch.zeros(N, device='cuda'), out=out, ldo=N, M=K, N=N, K=K)
    assert torch.equal(out, Wt.t().contiguous())


def test_gemm_concat_inplace_res(L):
    g = torch.Generator().manual_seed(5)
    M, K0, K1, N = 700, 256, 384, 256
    A0, A1 = torch.randn(M, K0, generator=g).cuda(), torch.randn(M, K1, generator=g).cuda()
    Wt = (torch.randn(N, K0 + K1, generator=g) / 25).cuda()
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device='cuda')
    _gemm(L, A0=A0, lda0=K0, K0=K0, A1=A1, lda1=K1, K1=K1, Wt=Wt, ldw=K0 + K1, bias=bias, out=out, ldo=N, M=M, N=N, K=K0 + K1)
    ref = torch.cat([A0, A1], 1).double() @ Wt.double().t() + bias.double()
    assert (out.double() - ref).abs().max().item() < 2e-5
    # in-place residual (fuse_feature_and_z): out aliases res
    f = torch.randn(M, N, generator=g).cuda()
    f0 = f.clone()
    z = torch.randn(M, 8, generator=g).cuda()
    Wz = torch.randn(N, 8, generator=g).cuda()
    _gemm(L, A0=z, lda0=8, K0=8, Wt=Wz, ldw=8, bias=bias, res=f, ldres=N, out=f, ldo=N, M=M, N=N, K=8, epi=3)
    ref = f0.double() + z.double() @ Wz.double().t() + bias.double()
    assert (f.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize('B,Ho,Wo,Cin,Cout', [(2, 6, 10, 192, 384), (1, 3, 5, 512, 512), (3, 1, 1, 384, 512)])
def test_gemm_patch2(L, B, Ho, Wo, Cin, Cout):
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, Cin, 2 * Ho, 2 * Wo, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 2, 2, generator=g) / (4 * Cin) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    out = torch.empty(B * Ho * Wo, Cout, device='cuda')
    _gemm(L, A0=xn, K0=Cin, H=Ho, W=Wo, Wt=wp, ldw=4 * Cin, bias=b, out=out, ldo=Cout, M=B * Ho * Wo, N=Cout, K=4 * Cin, a_mode=1)
    assert (out.view(B, Ho, Wo, Cout).double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('B,H,W,C,z', [(2, 5, 7, 256, 8), (1, 4, 6, 384, 96), (2, 2, 3, 512, 32), (1, 1, 1, 512, 32)])
def test_gemm_conv3(L, B, H, W, C, z):
    g = torch.Generator().manual_seed(C + z)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(z, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    b = torch.randn(z, generator=g).cuda()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(z, -1).contiguous()
    out = torch.empty(B * H * W, z, device='cuda')
    _gemm(L, A0=xn, K0=C, H=H, W=W, Wt=wp, ldw=9 * C, bias=b, out=out, ldo=z, M=B * H * W, N=z, K=9 * C, a_mode=2)
    assert (out.view(B, H, W, z).double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('B,H,W,Cin,Cout,r', [(2, 3, 5, 512, 384, 2), (1, 4, 4, 256, 128, 2)])
def test_gemm_pixel_shuffle(L, B, H, W, Cin, Cout, r):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout * r * r, Cin, 1, 1, generator=g) / Cin ** 0.5).cuda()
    b = torch.randn(Cout * r * r, generator=g).cuda()
    ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double()), r).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = w.reshape(Cout, r * r, Cin).permute(1, 0, 2).reshape(r * r * Cout, Cin).contiguous()
    bp = b.reshape(Cout, r * r).t().reshape(-1).contiguous()
    out = torch.full((B, H * r, W * r, Cout), float('nan'), device='cuda')
    _gemm(L, A0=xn, lda0=Cin, K0=Cin, H=H, W=W, Wt=wp, ldw=Cin, bias=bp, out=out, M=B * H * W, N=Cout * r * r, K=Cin, store=2, r=r)
    assert (out.double() - ref).abs().max().item() < 2e-5


def test_gemm_final_image(L):
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin = 2, 6, 9, 128
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(48, Cin, 1, 1, generator=g) / 4).cuda()
    b = torch.randn(48, generator=g).cuda()
    ref = F.pixel_shuffle(F.conv2d(x.double(), w.double(), b.double()), 4).clamp(-1, 1) * 0.5 + 0.5
    xn = x.permute(0, 2, 3, 1).contiguous()
    out = torch.full((B, 3, H * 4, W * 4), float('nan'), device='cuda')
    _gemm(L, A0=xn, lda0=Cin, K0=Cin, H=H, W=W, Wt=w.reshape(48, Cin).contiguous(), ldw=Cin, bias=b, out=out, M=B * H * W,
          N=48, K=Cin, store=3, r=4)
    assert (out.double() - ref).abs().max().item() < 2e-5


def test_gemm_batch_invariance(L):
    """Rows must be bit-identical whatever M / tile configuration: encoder (batched) and decoder (maybe not) must
    derive the same priors (SURVEY.md 'hard parts: enc/dec bit-consistency')."""
    g = torch.Generator().manual_seed(11)
    K, N = 512, 2048
    A = torch.randn(8 * 96, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    big = torch.empty(8 * 96, N, device='cuda')
    _gemm(L, A0=A, lda0=K, K0=K, Wt=Wt, ldw=K, bias=bias, out=big, ldo=N, M=8 * 96, N=N, K=K, epi=1)
    for i in (0, 5):
        small = torch.empty(96, N, device='cuda')
        _gemm(L, A0=A[i * 96:(i + 1) * 96].contiguous(), lda0=K, K0=K, Wt=Wt, ldw=K, bias=bias, out=small, ldo=N, M=96, N=N, K=K, epi=1)
        assert torch.equal(small, big[i * 96:(i + 1) * 96])
    A2 = torch.randn(24576, 192, generator=g).cuda()
    W2 = (torch.randn(384, 192, generator=g) / 14).cuda()
    b2 = torch.randn(384, generator=g).cuda()
    o1, o2 = torch.empty(24576, 384, device='cuda'), torch.empty(300, 384, device='cuda')
    _gemm(L, A0=A2, lda0=192, K0=192, Wt=W2, ldw=192, bias=b2, out=o1, ldo=384, M=24576, N=384, K=192)
    _gemm(L, A0=A2[5000:5300].contiguous(), lda0=192, K0=192, Wt=W2, ldw=192, bias=b2, out=o2, ldo=384, M=300, N=384, K=192)
    assert torch.equal(o2, o1[5000:5300])


@pytest.mark.parametrize('C,k', [(128, 7), (192, 7), (256, 7), (384, 5), (384, 7), (512, 1), (512, 3), (512, 5), (512, 7)])
@pytest.mark.parametrize('B,H,W', [(2, 9, 11), (1, 1, 1), (1, 2, 6)])
def test_dwconv_ln(L, C, k, B, H, W):
    g = torch.Generator().manual_seed(C * 10 + k + H)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(C, 1, k, k, generator=g) / k).cuda()
    b = torch.randn(C, generator=g).cuda()
    shift, scale = torch.randn(C, generator=g).cuda(), (0.3 * torch.randn(C, generator=g)).cuda()
    y = F.conv2d(x.double(), w.double(), b.double(), padding=(k - 1) // 2, groups=C).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), eps=1e-6) * (1 + scale.double()) + shift.double()
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = w.reshape(C, k * k).t().contiguous()
    out = torch.full((B, H, W, C), float('nan'), device='cuda')
    rc = L.lvae_dwconv_ln_f32(xn.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, shift.data_ptr(),
                              (1 + scale).contiguous().data_ptr(), out.data_ptr(), B, H, W, C, k, _st())
    assert rc == 0
    torch.cuda.synchronize()
    assert (out.double() - y).abs().max().item() < 3e-5
    # affine-LN variant (qres34m MyConvNeXtBlock)
    lw, lb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    y2 = F.conv2d(x.double(), w.double(), b.double(), padding=(k - 1) // 2, groups=C).permute(0, 2, 3, 1)
    y2 = F.layer_norm(y2, (C,), lw.double(), lb.double(), eps=1e-6)
    rc = L.lvae_dwconv_ln_f32(xn.data_ptr(), wp.data_ptr(), b.data_ptr(), lw.data_ptr(), lb.data_ptr(), None, None,
                              out.data_ptr(), B, H, W, C, k, _st())
    assert rc == 0
    torch.cuda.synchronize()
    assert (out.double() - y2).abs().max().item() < 3e-5


def test_dwconv_ln_beside_gemms(L):
    """The channel-per-lane depthwise+LN kernel on one stream while split-K bf16x3 GEMMs (MFMA) run on another: every output must
    equal the kernel's output when it runs alone.  Guards the packed-FMA operand form of dwconv_cl.hip: with the weight pair as
    src1 and `op_sel:[0,1,0]` the low lanes were occasionally wrong beside MFMA kernels (20-30 % of such launches on MI355X)."""
    import ctypes
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    g = torch.Generator().manual_seed(3)
    dw_cases = []
    for (B, H, W, C, k) in [(1, 16, 24, 384, 7), (1, 8, 12, 512, 3), (1, 32, 48, 256, 7), (1, 16, 24, 512, 5), (1, 32, 48, 192, 7)]:
        dw_cases.append(dict(B=B, H=H, W=W, C=C, k=k, x=torch.randn(B, H, W, C, generator=g).cuda(),
                             wp=(torch.randn(k * k, C, generator=g) / k).cuda(), b=torch.randn(C, generator=g).cuda(),
                             sh=torch.randn(C, generator=g).cuda(), sc=(1 + 0.3 * torch.randn(C, generator=g)).cuda()))

    def dw(c, y, st):
        assert L.lvae_dwconv_ln_f32(c['x'].data_ptr(), c['wp'].data_ptr(), c['b'].data_ptr(), None, None, c['sh'].data_ptr(),
                                    c['sc'].data_ptr(), y.data_ptr(), c['B'], c['H'], c['W'], c['C'], c['k'],
                                    ctypes.c_void_p(st.cuda_stream)) == 0

    gm_cases = []
    for (M, N, K, S) in [(384, 512, 1024, 8), (96, 1024, 512, 4), (384, 1536, 512, 4), (1536, 384, 768, 2), (384, 512, 1536, 4)]:
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        gm_cases.append(dict(M=M, N=N, K=K, S=S, A=torch.randn(M, K, generator=g).cuda(), Wt=Wt, W3=pack_bf16x3(Wt),
                             bias=torch.randn(N, generator=g).cuda()))
    ws = torch.empty(max(c['S'] * c['M'] * c['N'] for c in gm_cases), device='cuda')
    cnt = torch.zeros(8192, dtype=torch.int32, device='cuda')

    def gm(c, out, st):
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = c['A'].data_ptr(), c['K'], c['K'], c['Wt'].data_ptr(), c['W3'].data_ptr(), c['K']
        d.bias, d.out, d.ldo, d.M, d.N, d.K, d.epi, d.prec = c['bias'].data_ptr(), out.data_ptr(), c['N'], c['M'], c['N'], c['K'], 1, 2
        d.ksplit, d.ws, d.cnt = c['S'], ws.data_ptr(), cnt.data_ptr()
        assert L.lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(st.cuda_stream)) == 0

    cur = torch.cuda.current_stream()
    dw_ref, gm_ref = [], []
    for c in dw_cases:
        y = torch.empty_like(c['x']); dw(c, y, cur); torch.cuda.synchronize(); dw_ref.append(y)
    for c in gm_cases:
        o = torch.empty(c['M'], c['N'], device='cuda'); gm(c, o, cur); torch.cuda.synchronize(); gm_ref.append(o)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad_dw = bad_gm = 0
    notes = []
    for rep in range(40):
        outs_dw, outs_gm = [], []
        for i in range(10):
            di, gi = (i + rep) % len(dw_cases), (i + 2 * rep) % len(gm_cases)
            with torch.cuda.stream(s1):          # the NaN fill must be ordered before the kernel: same stream
                y = torch.full_like(dw_cases[di]['x'], float('nan'))
            dw(dw_cases[di], y, s1); outs_dw.append((y, dw_ref[di]))
            o = torch.empty(gm_cases[gi]['M'], gm_cases[gi]['N'], device='cuda')
            gm(gm_cases[gi], o, s2); outs_gm.append((o, gm_ref[gi]))
        torch.cuda.synchronize()
        bad_gm += sum(0 if torch.equal(a, b) else 1 for a, b in outs_gm)
        for a, b in outs_dw:
            if not torch.equal(a, b):
                bad_dw += 1
                if len(notes) < 6:          # which pixels, and do two fresh solo runs agree with the reference?
                    px = (a != b).any(dim=3).nonzero()
                    ci = next(i for i, r in enumerate(dw_ref) if r is b)
                    y1 = torch.empty_like(b); dw(dw_cases[ci], y1, cur); torch.cuda.synchronize()
                    notes.append(dict(shape=tuple(a.shape), n_px=len(px), first=px[:4].tolist(), max=float((a - b).abs().max()),
                                      solo_again_equals_ref=bool(torch.equal(y1, b)), solo_again_equals_this=bool(torch.equal(y1, a))))
    assert (bad_dw, bad_gm) == (0, 0), notes


def test_stem(L):
    g = torch.Generator().manual_seed(1)
    B, H, W, Cout = 2, 24, 40, 192
    im = torch.rand(B, 3, H, W, generator=g).cuda()
    w = (torch.randn(Cout, 3, 4, 4, generator=g) / 7).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    sh, sc = -0.4546259594901961, 3.67572653978347
    ref = F.conv2d((im.double() + sh) * sc, w.double(), b.double(), stride=4).permute(0, 2, 3, 1)
    out = torch.full((B, H // 4, W // 4, Cout), float('nan'), device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    wt = w.reshape(Cout, 48).t().contiguous()
    rc = L.lvae_stem_f32(im.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cout, sh, sc, flag.data_ptr(), _st())
    assert rc == 0
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() < 2e-5
    assert int(flag.item()) == 0                                   # every value in [0, 1]
    # the reference's input contract (qarv/model.py:219-220): out-of-range or NaN pixels raise the flag; NULL flag = unchecked
    for bad in (1.0000001, -1e-6, float('nan')):
        im2 = im.clone(); im2[1, 2, 5, 7] = bad
        flag.zero_()
        assert L.lvae_stem_f32(im2.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cout, sh, sc, flag.data_ptr(), _st()) == 0
        assert int(flag.item()) == 1, bad
        flag.zero_()
        assert L.lvae_range_flag_f32(im2.data_ptr(), im2.numel(), 0.0, 1.0, flag.data_ptr(), _st()) == 0
        assert int(flag.item()) == 1, bad
    flag.zero_()
    assert L.lvae_range_flag_f32(im.data_ptr(), im.numel(), 0.0, 1.0, flag.data_ptr(), _st()) == 0 and int(flag.item()) == 0
    assert L.lvae_stem_f32(im.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cout, sh, sc, None, _st()) == 0


def test_gemv(L):
    g = torch.Generator().manual_seed(2)
    N, K = 1000, 256
    Wt, b, x = torch.randn(N, K, generator=g).cuda() / 16, torch.randn(N, generator=g).cuda(), torch.randn(K, generator=g).cuda()
    y = torch.empty(N, device='cuda')
    for gi, go in ((0, 0), (1, 0), (0, 1)):
        assert L.lvae_gemv_f32(Wt.data_ptr(), b.data_ptr(), x.data_ptr(), y.data_ptr(), N, K, gi, go, _st()) == 0
        torch.cuda.synchronize()
        xin = F.gelu(x.double()) if gi else x.double()
        ref = Wt.double() @ xin + b.double()
        ref = F.gelu(ref) if go else ref
        assert (y.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize('B,HW,z,S', [(3, 35, 32, 4), (1, 96, 32, 4), (2, 1536, 96, 3), (4, 6144, 8, 2), (1, 70, 16, 7)])
def test_prior_index_behind_a_deferred_split_k_reduce(L, B, HW, z, S):
    """lvae_prior_index_sk_f32 (round 6): the prior parameters formed from the S partial-sum planes of a split-K GEMM whose reduce pass was
    deferred -- ((ws[0] + ws[1]) + ...) + bias in slice order -- then indexed, against the two-launch form: the sum done by torch in the
    same order, then lvae_prior_index_f32.  prm, pm and every index bit-equal; and end to end against lvae_gemm_f32 itself: the same GEMM
    with and without `defer_reduce` (the latter runs the library's own reduce pass)."""
    import ctypes
    from lvae import _native
    from lvae.models.base import pack_f16x2
    g = torch.Generator().manual_seed(B * HW + z + S)
    M, N = B * HW, 2 * z
    ws = (torch.randn(S, M, N, generator=g) * torch.exp(torch.randn(S, 1, 1, generator=g))).cuda()
    bias = torch.randn(N, generator=g).cuda()
    table = torch.exp(torch.linspace(np.log(0.11), np.log(20.0), 64)).cuda()
    ref_prm = ws[0].clone()
    for sl in range(1, S):
        ref_prm = ref_prm + ws[sl]
    ref_prm = ref_prm + bias
    pm0, idx0 = torch.empty(M, z, device='cuda'), torch.empty(B, z, HW, dtype=torch.uint8, device='cuda')
    assert L.lvae_prior_index_f32(ref_prm.data_ptr(), pm0.data_ptr(), idx0.data_ptr(), table.data_ptr(), 64, float(table[0]), B, HW, z, None, _st()) == 0
    prm = torch.full((M, N), float('nan'), device='cuda')
    pm1, idx1 = torch.empty(M, z, device='cuda'), torch.empty(B, z, HW, dtype=torch.uint8, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert L.lvae_prior_index_sk_f32(ws.data_ptr(), S, bias.data_ptr(), prm.data_ptr(), pm1.data_ptr(), idx1.data_ptr(), table.data_ptr(), 64,
                                     float(table[0]), B, HW, z, flag.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(prm, ref_prm) and torch.equal(pm1, pm0) and torch.equal(idx1, idx0) and int(flag) == 0
    # the GEMM itself, K = 512 in S slices: deferred planes + this kernel == the GEMM's own reduce pass + lvae_prior_index_f32
    if 512 % (32 * S) == 0:
        K = 512
        A = torch.randn(M, K, generator=g).cuda()
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        w16 = pack_f16x2(Wt)

        def desc(defer, out, wsb):
            d = _native.GemmDesc()
            d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), K, K, Wt.data_ptr(), w16.data_ptr(), K
            d.bias, d.out, d.ldo, d.M, d.N, d.K, d.epi, d.prec = bias.data_ptr(), out.data_ptr(), N, M, N, K, 0, 4
            d.ksplit, d.ws, d.defer_reduce = S, wsb.data_ptr(), defer
            return d
        out_a = torch.full((M, N), float('nan'), device='cuda'); ws_a = torch.empty(S * M * N, device='cuda')
        out_b = torch.full((M, N), float('nan'), device='cuda'); ws_b = torch.full((S * M * N,), float('nan'), device='cuda')
        assert L.lvae_gemm_f32(ctypes.byref(desc(0, out_a, ws_a)), _st()) == 0
        assert L.lvae_gemm_f32(ctypes.byref(desc(1, out_b, ws_b)), _st()) == 0
        torch.cuda.synchronize()
        assert torch.isnan(out_b).all() and not torch.isnan(ws_b).any()          # deferred: planes written, no reduce pass
        assert L.lvae_prior_index_f32(out_a.data_ptr(), pm0.data_ptr(), idx0.data_ptr(), table.data_ptr(), 64, float(table[0]), B, HW, z, None, _st()) == 0
        assert L.lvae_prior_index_sk_f32(ws_b.data_ptr(), S, bias.data_ptr(), prm.data_ptr(), pm1.data_ptr(), idx1.data_ptr(), table.data_ptr(), 64,
                                         float(table[0]), B, HW, z, None, _st()) == 0
        torch.cuda.synchronize()
        assert torch.equal(prm, out_a) and torch.equal(pm1, pm0) and torch.equal(idx1, idx0)


@pytest.mark.parametrize('B,HW,z,ldz,S', [(3, 35, 32, 32, 4), (1, 96, 32, 32, 36), (2, 1536, 96, 96, 9), (4, 6144, 8, 32, 4), (1, 70, 16, 32, 7)])
def test_quantize_behind_a_deferred_split_k_reduce(L, B, HW, z, ldz, S):
    """lvae_quantize_sk_f32 (round 6): the posterior mean formed from the S planes of a split-K GEMM whose reduce pass was deferred, then
    quantised, against the two-launch form (the sum done by torch in slice order, then lvae_quantize_f32): qm, symbols and zhat bit-equal,
    padded zhat columns zeroed, and the non-finite flag raised by a NaN that only appears in the sum."""
    g = torch.Generator().manual_seed(B * HW + z + S)
    M = B * HW
    ws = (torch.randn(S, M, z, generator=g) * 3.0).cuda()
    bias = torch.randn(z, generator=g).cuda()
    pm = torch.randn(M, z, generator=g).cuda()
    ref_qm = ws[0].clone()
    for sl in range(1, S):
        ref_qm = ref_qm + ws[sl]
    ref_qm = ref_qm + bias
    sym0 = torch.empty(B, z, HW, dtype=torch.int32, device='cuda'); zh0 = torch.full((M, ldz), float('nan'), device='cuda')
    assert L.lvae_quantize_f32(ref_qm.data_ptr(), pm.data_ptr(), sym0.data_ptr(), zh0.data_ptr(), B, HW, z, ldz, None, _st()) == 0
    qm = torch.full((M, z), float('nan'), device='cuda')
    sym1 = torch.empty(B, z, HW, dtype=torch.int32, device='cuda'); zh1 = torch.full((M, ldz), float('nan'), device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert L.lvae_quantize_sk_f32(ws.data_ptr(), S, bias.data_ptr(), qm.data_ptr(), pm.data_ptr(), sym1.data_ptr(), zh1.data_ptr(), B, HW, z, ldz,
                                  flag.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(qm, ref_qm) and torch.equal(sym1, sym0) and torch.equal(zh1, zh0) and int(flag) == 0
    assert (zh1[:, z:] == 0).all()
    ws[0, 0, 0], ws[S - 1, 0, 0] = float('inf'), float('-inf')              # finite planes apart, NaN in the sum
    assert L.lvae_quantize_sk_f32(ws.data_ptr(), S, bias.data_ptr(), qm.data_ptr(), pm.data_ptr(), sym1.data_ptr(), zh1.data_ptr(), B, HW, z, ldz,
                                  flag.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    assert int(flag) != 0 and torch.isnan(qm[0, 0])
    assert L.lvae_quantize_sk_f32(ws.data_ptr(), 1, bias.data_ptr(), qm.data_ptr(), pm.data_ptr(), sym1.data_ptr(), zh1.data_ptr(), B, HW, z, ldz,
                                  None, _st()) == -22


def test_prior_index_quantize_dequantize(L):
    """Integer outputs must be EXACT against the reference formulation computed by torch on the same device values
    away from decision boundaries, and against an fp64 evaluation elsewhere."""
    g = torch.Generator().manual_seed(4)
    B, HW, z = 3, 35, 32
    M = B * HW
    prm = (torch.randn(M, 2 * z, generator=g) * 2).cuda()
    table = torch.exp(torch.linspace(np.log(0.11), np.log(20.0), 64)).cuda()
    pm, idx = torch.empty(M, z, device='cuda'), torch.empty(B, z, HW, dtype=torch.uint8, device='cuda')
    assert L.lvae_prior_index_f32(prm.data_ptr(), pm.data_ptr(), idx.data_ptr(), table.data_ptr(), 64, float(table[0]), B, HW, z, None, _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(pm, prm[:, :z])
    pv64 = torch.exp(F.softplus(prm[:, z:].double() + 2.3) - 2.3)
    s64 = torch.clamp(pv64, min=float(table[0]))
    ref = (table.double()[None, None, :63] < s64[..., None]).sum(-1)            # #{i<63: table[i] < s}
    ref = ref.view(B, HW, z).permute(0, 2, 1)
    near = ((s64[..., None] / table.double()[None, None, :] - 1).abs().min(-1)[0] < 1e-5).view(B, HW, z).permute(0, 2, 1)
    assert torch.equal(idx.long()[~near], ref[~near])
    # quantize / dequantize
    qm = (torch.randn(M, z, generator=g) * 6).cuda()
    sym, zhat = torch.empty(B, z, HW, dtype=torch.int32, device='cuda'), torch.empty(M, z, device='cuda')
    assert L.lvae_quantize_f32(qm.data_ptr(), pm.data_ptr(), sym.data_ptr(), zhat.data_ptr(), B, HW, z, z, None, _st()) == 0
    torch.cuda.synchronize()
    r = torch.round(qm - pm)
    assert torch.equal(sym, r.int().view(B, HW, z).permute(0, 2, 1).contiguous())
    assert torch.equal(zhat, r + pm)
    z2 = torch.empty(M, z, device='cuda')
    assert L.lvae_dequantize_f32(sym.data_ptr(), pm.data_ptr(), z2.data_ptr(), B, HW, z, z, _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(z2, zhat)
    # half-way cases: round-half-to-even
    qh = torch.tensor([[0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, 4.5]], device='cuda')
    ph = torch.zeros_like(qh)
    sh, zh = torch.empty(1, 8, 1, dtype=torch.int32, device='cuda'), torch.empty(1, 8, device='cuda')
    assert L.lvae_quantize_f32(qh.data_ptr(), ph.data_ptr(), sh.data_ptr(), zh.data_ptr(), 1, 1, 8, 8, None, _st()) == 0
    torch.cuda.synchronize()
    assert sh.flatten().tolist() == [0, 2, 2, 0, -2, -2, 4, 4]


@pytest.mark.parametrize('B,HW,z,ldz,pinned', [(2, 200, 96, 96, False), (3, 64, 70, 72, False), (1, 129, 8, 8, True), (2, 96, 130, 132, True)])
def test_index_quantize_tiles_and_pinned_rasters(L, B, HW, z, ldz, pinned):
    """The raster-side kernels work on tiles of 64 pixels x 64 channels (csrc/pointwise.hip): several pixel tiles, ragged last tile,
    more than one channel chunk, padded rows -- and, `pinned`, with the NCHW rasters in pinned HOST memory (what the native group loops
    hand them: lvae_dec_block.idx_dev = NULL), read back on the host after a synchronisation."""
    g = torch.Generator().manual_seed(HW + z)
    M = B * HW
    prm = (torch.randn(M, 2 * z, generator=g) * 2).cuda()
    table = torch.exp(torch.linspace(np.log(0.11), np.log(20.0), 64)).cuda()
    pm = torch.empty(M, z, device='cuda')
    raster = (lambda dt: torch.empty(B, z, HW, dtype=dt).pin_memory()) if pinned else (lambda dt: torch.empty(B, z, HW, dtype=dt, device='cuda'))
    idx, sym = raster(torch.uint8), raster(torch.int32)
    assert L.lvae_prior_index_f32(prm.data_ptr(), pm.data_ptr(), idx.data_ptr(), table.data_ptr(), 64, float(table[0]), B, HW, z, None, _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(pm, prm[:, :z])
    s64 = torch.clamp(torch.exp(F.softplus(prm[:, z:].double() + 2.3) - 2.3), min=float(table[0]))
    ref = (table.double()[None, None, :63] < s64[..., None]).sum(-1).view(B, HW, z).permute(0, 2, 1)
    near = ((s64[..., None] / table.double()[None, None, :] - 1).abs().min(-1)[0] < 1e-5).view(B, HW, z).permute(0, 2, 1)
    assert torch.equal(idx.cuda().long()[~near], ref[~near])
    qm = (torch.randn(M, z, generator=g) * 6).cuda()
    zhat = torch.full((M, ldz), float('nan'), device='cuda')
    assert L.lvae_quantize_f32(qm.data_ptr(), pm.data_ptr(), sym.data_ptr(), zhat.data_ptr(), B, HW, z, ldz, None, _st()) == 0
    torch.cuda.synchronize()
    r = torch.round(qm - pm)
    assert torch.equal(sym.cuda(), r.int().view(B, HW, z).permute(0, 2, 1).contiguous())
    assert torch.equal(zhat[:, :z], r + pm) and not zhat[:, z:].any()
    if pinned:                                   # the decode direction: the HOST writes the symbols, the kernel reads them in place
        sym.copy_(torch.randint(-40, 41, (B, z, HW), generator=g, dtype=torch.int32))
    z2 = torch.full((M, ldz), float('nan'), device='cuda')
    assert L.lvae_dequantize_f32(sym.data_ptr(), pm.data_ptr(), z2.data_ptr(), B, HW, z, ldz, _st()) == 0
    torch.cuda.synchronize()
    want = sym.cuda().permute(0, 2, 1).reshape(M, z).float() + pm
    assert torch.equal(z2[:, :z], want) and not z2[:, z:].any()


def test_sqerr(L):
    a, b = torch.rand(2, 3 * 70 * 90, device='cuda'), torch.rand(2, 3 * 70 * 90, device='cuda')
    out = torch.zeros(2, dtype=torch.float64, device='cuda')
    assert L.lvae_sqerr_sum_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2, a.shape[1], _st()) == 0
    torch.cuda.synchronize()
    ref = (a.double() - b.double()).square().sum(1)
    assert torch.allclose(out, ref, rtol=1e-6)   # kernel forms a-b in fp32 (as the reference does) then accumulates in fp64


def test_gelu_erf_accuracy(L):
    """The fused epilogues' exact-erf GELU (csrc/device_math.h) vs fp64: |err| <= 2e-7 * max(1,|x|) on a dense grid."""
    x = torch.cat([torch.linspace(-8, 8, 400001), torch.linspace(-1.5, 1.5, 200001)]).cuda()
    y = torch.empty_like(x)
    assert L.lvae_gelu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _st()) == 0
    torch.cuda.synchronize()
    ref = F.gelu(x.double())
    err = (y.double() - ref).abs() / torch.clamp(x.double().abs(), min=1.0)
    assert float(err.max()) <= 2e-7, float(err.max())


def test_quantize_padded_rows_and_gemm_a_gelu(L):
    """qres34m pieces: zhat rows padded to a multiple of 4 channels (z = 14 -> 16), and GELU applied to A on load."""
    g = torch.Generator().manual_seed(8)
    B, HW, z, ldz = 2, 15, 14, 16
    M = B * HW
    qm, pm = (torch.randn(M, z, generator=g) * 5).cuda(), torch.randn(M, z, generator=g).cuda()
    sym = torch.empty(B, z, HW, dtype=torch.int32, device='cuda')
    zhat = torch.full((M, ldz), float('nan'), device='cuda')
    assert L.lvae_quantize_f32(qm.data_ptr(), pm.data_ptr(), sym.data_ptr(), zhat.data_ptr(), B, HW, z, ldz, None, _st()) == 0
    torch.cuda.synchronize()
    r = torch.round(qm - pm)
    assert torch.equal(zhat[:, :z], r + pm) and torch.equal(zhat[:, z:], torch.zeros(M, 2, device='cuda'))
    z2 = torch.full((M, ldz), float('nan'), device='cuda')
    assert L.lvae_dequantize_f32(sym.data_ptr(), pm.data_ptr(), z2.data_ptr(), B, HW, z, ldz, _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(z2, zhat)
    # c1(gelu(cat[f, e])): two A sources + GELU on load, N = 96, then a 3x3 over 96 channels with GELU epilogue
    Mx, K0, K1, N = 300, 384, 384, 96
    A0, A1 = torch.randn(Mx, K0, generator=g).cuda(), torch.randn(Mx, K1, generator=g).cuda()
    Wt = (torch.randn(N, K0 + K1, generator=g) / 27).cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(Mx, N, device='cuda')
    _gemm(L, A0=A0, lda0=K0, K0=K0, A1=A1, lda1=K1, K1=K1, Wt=Wt, ldw=K0 + K1, bias=b, out=out, ldo=N, M=Mx, N=N, K=K0 + K1, a_gelu=1)
    ref = F.gelu(torch.cat([A0, A1], 1).double()) @ Wt.double().t() + b.double()
    assert (out.double() - ref).abs().max().item() < 2e-5
    Bc, H, W, C = 2, 5, 6, 48
    x = torch.randn(Bc, C, H, W, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) / 20).cuda()
    bb = torch.randn(C, generator=g).cuda()
    ref = F.gelu(F.conv2d(x.double(), w.double(), bb.double(), padding=1)).permute(0, 2, 3, 1)
    o2 = torch.empty(Bc * H * W, C, device='cuda')
    _gemm(L, A0=x.permute(0, 2, 3, 1).contiguous(), K0=C, H=H, W=W, Wt=w.permute(0, 2, 3, 1).reshape(C, -1).contiguous(), ldw=9 * C,
          bias=bb, out=o2, ldo=C, M=Bc * H * W, N=C, K=9 * C, a_mode=2, epi=1)
    assert (o2.view(Bc, H, W, C).double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize('prec', [0, 2])
@pytest.mark.parametrize('M,N,K,S,epi,a_mode', [(384, 512, 1024, 8, 2, 0), (96, 512, 2048, 8, 1, 0), (768, 1024, 512, 4, 1, 0),
                                                (300, 32, 4608, 16, 0, 2), (130, 64, 1152, 9, 3, 0)])
def test_gemm_split_k(M, N, K, S, epi, a_mode, prec):
    """Split-K (ksplit = S): fp32-class accuracy against an fp64 reference, deterministic, and -- the property the host relies on --
    independent of M for a fixed S: the rows of a batch are bit-identical to the rows of a smaller call."""
    import ctypes
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(M + N + K + S)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if a_mode == 2:                      # 3x3 gather over an NHWC map: M = B*H*W pixels, K = 9*Cin
        Hh, Ww, Cin = 10, M // 10, K // 9
        A = torch.randn(M, Cin, generator=g).cuda()
    else:
        Hh = Ww = 0
        Cin = K
        A = torch.randn(M, K, generator=g).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    W3 = pack_bf16x3(Wt)
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()

    cnt = torch.zeros(4096, dtype=torch.int32, device='cuda')       # arrival counters of the in-kernel reduction (zero at rest)

    def run(m_rows, ksplit, in_kernel=True, reps=1):
        out = torch.full((m_rows, N), float('nan'), device='cuda')
        ws = torch.full((max(1, ksplit) * m_rows * N,), float('nan'), device='cuda')
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), Cin, Cin, Wt.data_ptr(), W3.data_ptr(), K
        d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec, d.a_mode, d.H, d.W = m_rows, N, K, epi, prec, a_mode, Hh, Ww
        d.ksplit, d.ws = ksplit, ws.data_ptr()
        d.cnt = cnt.data_ptr() if (in_kernel and ksplit > 1) else None
        for _ in range(reps):
            assert L.lvae_gemm_f32(ctypes.byref(d), st) == 0
        torch.cuda.synchronize()
        return out

    o1, oS, oS2 = run(M, 1), run(M, S), run(M, S, in_kernel=False)
    # the in-kernel reduction (a tile's last-arriving slice workgroup sums the slabs in slice order) == the two-kernel form, bit for
    # bit, and leaves its arrival counters zero for the next launch; repeated back-to-back launches reuse workspace and counters
    assert torch.equal(oS, oS2)
    assert int(cnt.abs().sum().item()) == 0
    assert torch.equal(run(M, S, reps=5), oS) and int(cnt.abs().sum().item()) == 0
    if a_mode == 0:
        ref = A.double() @ Wt.double().t() + bias.double()
        ref = {0: ref, 1: torch.nn.functional.gelu(ref), 2: res.double() + gamma.double() * ref, 3: res.double() + ref}[epi]
        e1, eS = float((o1.double() - ref).abs().max()), float((oS.double() - ref).abs().max())
        assert eS <= 2 * e1 + 2e-6, (e1, eS)
        half = (M // 2) // 2 * 2
        assert torch.equal(run(half, S), oS[:half])          # M-independence (rows of a smaller call)
    else:
        assert float((oS - o1).abs().max()) <= 2e-5 * max(1.0, float(o1.abs().max()))


def test_gemm_split_k_in_kernel_reduction_under_load():
    """The hand-off inside the launch (slab stores -> agent-scope release -> ticket -> last arriver's agent-scope acquire -> slab
    loads) must hold under UNEVEN load with the consumer's caches warm (MI355X_MICROARCH.md, visibility rules): many different
    shapes back to back on two streams sharing the GPU, every output word compared with the two-kernel form."""
    import ctypes
    from lvae import _native
    from lvae.models.base import pack_bf16x3
    L = _native.lib()
    g = torch.Generator().manual_seed(7)
    cases = []
    for (M, N, K, S) in [(3072, 512, 1024, 2), (768, 1024, 512, 4), (3072, 512, 1536, 4), (96, 512, 2048, 16), (1536, 384, 768, 2),
                         (384, 512, 1024, 8), (12288, 96, 3456, 4)]:
        A = torch.randn(M, K, generator=g).cuda()
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        cases.append(dict(M=M, N=N, K=K, S=S, A=A, Wt=Wt, W3=pack_bf16x3(Wt), bias=torch.randn(N, generator=g).cuda(),
                          res=torch.randn(M, N, generator=g).cuda(), gamma=torch.rand(N, generator=g).cuda()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def launch(c, out, ws, cnt, stream, in_kernel):
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = c['A'].data_ptr(), c['K'], c['K'], c['Wt'].data_ptr(), c['W3'].data_ptr(), c['K']
        d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = c['bias'].data_ptr(), c['gamma'].data_ptr(), c['res'].data_ptr(), c['N'], out.data_ptr(), c['N']
        d.M, d.N, d.K, d.epi, d.prec = c['M'], c['N'], c['K'], 2, 2
        d.ksplit, d.ws, d.cnt = c['S'], ws.data_ptr(), (cnt.data_ptr() if in_kernel else None)
        assert L.lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(stream.cuda_stream)) == 0

    refs = []
    for c in cases:
        out = torch.empty(c['M'], c['N'], device='cuda'); ws = torch.empty(c['S'] * c['M'] * c['N'], device='cuda')
        launch(c, out, ws, None, torch.cuda.current_stream(), False)
        torch.cuda.synchronize()
        refs.append(out)
    bufs = []
    for si, stq in enumerate(streams):      # per stream: its own workspace + counters (as the two pipeline groups have), shared by all shapes
        ws = torch.empty(max(c['S'] * c['M'] * c['N'] for c in cases), device='cuda')
        bufs.append((ws, torch.zeros(8192, dtype=torch.int32, device='cuda')))
    torch.cuda.synchronize()
    bad = 0
    for rep in range(6):
        outs = []
        for i in range(len(cases) * 2):
            si = i % 2
            ci = (i // 2 + rep * (si + 1)) % len(cases)
            c = cases[ci]
            out = torch.empty(c['M'], c['N'], device='cuda')
            launch(c, out, bufs[si][0], bufs[si][1], streams[si], True)
            outs.append((out, refs[ci]))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, r) else 1 for o, r in outs)
    assert bad == 0
    assert all(int(b[1].abs().sum().item()) == 0 for b in bufs)


@pytest.mark.parametrize('C,k,H,W,B', [(192, 7, 96, 160, 7), (128, 7, 61, 99, 18), (192, 5, 70, 130, 11), (128, 5, 64, 128, 12),
                                       (128, 7, 17, 33, 170)])
def test_dwconv_ln_large_map_variants_same_bits(L, C, k, H, W, B):
    """The depthwise+LN launcher picks its tile height (rows per workgroup) from the map size and the batch: a large batch must give,
    image by image, the bits of single-image calls -- ragged sizes included."""
    g = torch.Generator().manual_seed(C + k + H + W)
    x1 = torch.randn(3, H, W, C, generator=g).cuda()
    x = x1.repeat((B + 2) // 3, 1, 1, 1)[:B].contiguous()
    wp = (torch.randn(k * k, C, generator=g) / k).cuda()
    b = torch.randn(C, generator=g).cuda()
    shift, sc1 = torch.randn(C, generator=g).cuda(), (1 + 0.3 * torch.randn(C, generator=g)).cuda()
    assert B * H * W >= 90000 > H * W
    out = torch.full((B, H, W, C), float('nan'), device='cuda')
    assert L.lvae_dwconv_ln_f32(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, shift.data_ptr(), sc1.data_ptr(), out.data_ptr(),
                                B, H, W, C, k, _st()) == 0
    singles = []
    for i in range(3):
        o = torch.full((1, H, W, C), float('nan'), device='cuda')
        xi = x1[i:i + 1].contiguous()
        assert L.lvae_dwconv_ln_f32(xi.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, shift.data_ptr(), sc1.data_ptr(), o.data_ptr(),
                                    1, H, W, C, k, _st()) == 0
        singles.append(o)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    for i in range(B):
        assert torch.equal(out[i:i + 1], singles[i % 3]), i
