"""Debug: one ConvNeXt block in the fp8 mode, Q8 pipeline vs in-GEMM quantiser vs fp64, on the model's (C, hid, k) shapes."""
import ctypes, os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch, torch.nn.functional as F
from lvae import _native
from lvae.models.base import pack_mxfp8, pack_mxfp8_q8, unpack_mxfp8_q8
L = _native.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def gemm(**kw):
    d = _native.GemmDesc(); keep = []
    for k, v in kw.items():
        if torch.is_tensor(v): keep.append(v); v = v.data_ptr()
        setattr(d, k, v)
    d.prec = 3
    rc = L.lvae_gemm_f32(ctypes.byref(d), st()); assert rc == 0, rc
    torch.cuda.synchronize()
g = torch.Generator().manual_seed(0)
for (B, H, W, C, hid, k) in [(2, 32, 48, 128, 192, 7), (2, 16, 24, 256, 448, 7), (2, 32, 48, 192, 384, 7), (2, 8, 12, 384, 768, 5), (1, 19, 19, 512, 2048, 1)]:
    M = B * H * W
    x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16).cuda()
    wt = (torch.randn(k * k, C, generator=g) / k).cuda(); bias = torch.randn(C, generator=g).cuda()
    sh, sc = torch.randn(C, generator=g).cuda(), (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    W1 = (torch.randn(hid, C, generator=g) / C ** 0.5); b1 = torch.randn(hid, generator=g).cuda()
    W2 = (torch.randn(C, hid, generator=g) / hid ** 0.5); b2 = torch.randn(C, generator=g).cuda()
    gam = torch.rand(C, generator=g).cuda()
    # fp64 reference of the block (no quantisation)
    yf = torch.empty(B, H, W, C, device='cuda')
    assert L.lvae_dwconv_ln_f32(x.float().data_ptr(), wt.data_ptr(), bias.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), yf.data_ptr(), B, H, W, C, k, st()) == 0
    ref = x.double().view(M, C) + gam.double() * (F.gelu(yf.double().view(M, C) @ W1.double().cuda().t() + b1.double()) @ W2.double().cuda().t() + b2.double())
    # in-GEMM quantiser
    y = torch.empty(B, H, W, C, device='cuda', dtype=torch.bfloat16)
    assert L.lvae_dwconv_ln_bf16(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), y.data_ptr(), B, H, W, C, k, st()) == 0
    h = torch.empty(M, hid, device='cuda', dtype=torch.bfloat16); o1 = torch.empty(M, C, device='cuda', dtype=torch.bfloat16)
    gemm(A0=y, lda0=C, K0=C, Wt16=pack_mxfp8(W1).cuda(), ldw=(C + 63) // 64 * 64, bias=b1, out=h, ldo=hid, M=M, N=hid, K=C, epi=1, a_bf16=1, out_bf16=1)
    gemm(A0=h, lda0=hid, K0=hid, Wt16=pack_mxfp8(W2).cuda(), ldw=(hid + 63) // 64 * 64, bias=b2, gamma=gam, res=x, ldres=C, out=o1, ldo=C, M=M, N=C, K=hid, epi=2, a_bf16=1, out_bf16=1)
    # Q8 pipeline
    yq = torch.zeros(M * C * 2, device='cuda', dtype=torch.uint8)
    assert L.lvae_dwconv_ln_q8(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), yq.data_ptr(), B, H, W, C, k, st()) == 0
    hq = torch.zeros(M * hid * 2, device='cuda', dtype=torch.uint8); o2 = torch.empty(M, C, device='cuda', dtype=torch.bfloat16)
    gemm(A0=yq, lda0=C, K0=C, Wt16=pack_mxfp8_q8(W1).cuda(), ldw=C, bias=b1, out=hq, ldo=hid, M=M, N=hid, K=C, epi=1, a_h2=1, out_h2=1, out_bf16=1)
    gemm(A0=hq, lda0=hid, K0=hid, Wt16=pack_mxfp8_q8(W2).cuda(), ldw=hid, bias=b2, gamma=gam, res=x, ldres=C, out=o2, ldo=C, M=M, N=C, K=hid, epi=2, a_h2=1, out_bf16=1)
    # fc1 again from a HOST-packed copy of the same A values, and into an exactly-sized output buffer
    Aq2 = pack_mxfp8_q8(unpack_mxfp8_q8(yq.cpu()[:M * C + M * C // 32], M, C)).cuda()
    same_bytes = torch.equal(Aq2.cpu(), yq.cpu()[:M * C + M * C // 32])
    hq2 = torch.zeros(M * hid + M * hid // 32, device='cuda', dtype=torch.uint8)
    gemm(A0=Aq2, lda0=C, K0=C, Wt16=pack_mxfp8_q8(W1).cuda(), ldw=C, bias=b1, out=hq2, ldo=hid, M=M, N=hid, K=C, epi=1, a_h2=1, out_h2=1)
    hd2 = unpack_mxfp8_q8(hq2.cpu(), M, hid)
    yd = unpack_mxfp8_q8(yq.cpu()[:M * C + M * C // 32], M, C); hd = unpack_mxfp8_q8(hq.cpu()[:M * hid + M * hid // 32], M, hid)
    href = F.gelu(yd.double().cuda() @ unpack_mxfp8_q8(pack_mxfp8_q8(W1), hid, C).double().cuda().t() + b1.double())
    E = (hd2.cuda().double() - href)
    print('   err rms per 64-column tile:', [round(float(E[:, c:c + 64].square().mean().sqrt()), 3) for c in range(0, hid, 64)][:12],
          ' per 32-row block (first 8):', [round(float(E[r:r + 32].square().mean().sqrt()), 3) for r in range(0, min(M, 256), 32)])
    # without GELU / bias: raw products
    hq3 = torch.zeros(M * hid + M * hid // 32, device='cuda', dtype=torch.uint8)
    gemm(A0=Aq2, lda0=C, K0=C, Wt16=pack_mxfp8_q8(W1).cuda(), ldw=C, out=hq3, ldo=hid, M=M, N=hid, K=C, epi=0, a_h2=1, out_h2=1)
    P = yd.double().cuda() @ unpack_mxfp8_q8(pack_mxfp8_q8(W1), hid, C).double().cuda().t()
    ob = torch.empty(M, hid, device='cuda', dtype=torch.bfloat16)
    gemm(A0=Aq2, lda0=C, K0=C, Wt16=pack_mxfp8_q8(W1).cuda(), ldw=C, out=ob, ldo=hid, M=M, N=hid, K=C, epi=0, a_h2=1, out_bf16=1)
    print('   raw product: q8-out rms err', float((unpack_mxfp8_q8(hq3.cpu(), M, hid).cuda().double() - P).square().mean().sqrt()), ' bf16-out rms err',
          float((ob.double() - P).square().mean().sqrt()), ' rms P', float(P.square().mean().sqrt()))
    e1, e2 = (o1.double() - ref), (o2.double() - ref)
    print(f'C={C} hid={hid} k={k} M={M}: block rms error vs fp64: in-GEMM {float(e1.square().mean().sqrt()):.4e}  Q8 {float(e2.square().mean().sqrt()):.4e}'
          f'   | y: rms(q8 - fp32) {float((yd.cuda() - yf.view(M, C)).square().mean().sqrt()):.3e}  h: rms(q8 - its fp64) {float((hd.cuda().double() - href).square().mean().sqrt()):.3e}; host-packed A same bytes {same_bytes}, h from it: rms err {float((hd2.cuda().double() - href).square().mean().sqrt()):.3e}; rms h {float(href.square().mean().sqrt()):.3f} rms hd {float(hd.square().mean().sqrt()):.3f}', flush=True)
