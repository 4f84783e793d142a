"""Round 6: lvae_mlp_sk (fused small-map MLP + reduce launch) against the launches it replaces, per shape: python tools/r6_mlp_sk_bench.py"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_f16x2_k32, pack_f16x2
L = _native.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def gemm(A, K, Wt, W16, bias, out, N, M, epi, **kw):
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = A.data_ptr(), K, K, Wt.data_ptr(), W16.data_ptr(), K
    d.bias, d.out, d.ldo, d.M, d.N, d.K, d.epi, d.prec = bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, 4
    for k, v in kw.items(): setattr(d, k, v)
    return d
def timeit(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f'{"M":>6} {"C":>4} {"hid":>5} {"S1":>3} {"S2":>3} | serial fc1+fc2 | parallel fc1+fc2(+reduce) | mlp_sk (fused + reduce)   [us]')
for (M, C, HID, S1, S2) in [(96, 512, 2048, 4, 16), (384, 512, 2048, 4, 16), (768, 512, 2048, 4, 16), (96, 512, 1024, 4, 8), (384, 512, 1024, 4, 8), (768, 512, 1024, 4, 8),
                            (384, 512, 1536, 2, 8), (1536, 512, 1536, 2, 8), (3072, 512, 1536, 2, 8), (384, 512, 1024, 4, 8), (1536, 512, 1024, 4, 8), (3072, 512, 1024, 4, 8)]:
    g = torch.Generator().manual_seed(1)
    yf = torch.randn(M, C, generator=g).cuda(); W1 = (torch.randn(HID, C, generator=g) / C ** 0.5).cuda(); W2 = (torch.randn(C, HID, generator=g) / HID ** 0.5).cuda()
    b1, b2, gamma = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda()
    res = torch.randn(M, C, generator=g).cuda()
    y, w1h, w2h, w1p, w2p = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2), pack_f16x2(W1), pack_f16x2(W2)
    hid = torch.empty(M, HID, device='cuda'); out = torch.empty(M, C, device='cuda'); ws = torch.empty(max(S1 * M * HID, S2 * M * C), device='cuda')
    d1 = gemm(y, C, W1, w1h, b1, hid, HID, M, 1, a_h2=1, out_h2=1, ksplit=S1, ws=ws.data_ptr())
    d2 = gemm(hid, HID, W2, w2h, b2, out, C, M, 2, gamma=gamma.data_ptr(), res=res.data_ptr(), ldres=C, a_h2=1, ksplit=S2, ws=ws.data_ptr())
    t_ser = timeit(lambda: (L.lvae_gemm_f32(ctypes.byref(d1), st()), L.lvae_gemm_f32(ctypes.byref(d2), st())))
    p1 = gemm(yf, C, W1, w1p, b1, hid, HID, M, 1, ksplit=S1, ws=ws.data_ptr())
    p2 = gemm(hid, HID, W2, w2p, b2, out, C, M, 2, gamma=gamma.data_ptr(), res=res.data_ptr(), ldres=C, ksplit=S2, ws=ws.data_ptr())
    t_par = timeit(lambda: (L.lvae_gemm_f32(ctypes.byref(p1), st()), L.lvae_gemm_f32(ctypes.byref(p2), st())))
    d = _native.MlpSkDesc()
    d.y, d.w1, d.b1, d.w2, d.b2, d.gamma = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr()
    d.res, d.out, d.ws, d.M, d.C, d.hid, d.S1, d.S2 = res.data_ptr(), out.data_ptr(), ws.data_ptr(), M, C, HID, S1, S2
    t_sk = timeit(lambda: L.lvae_mlp_sk(ctypes.byref(d), st()))
    print(f'{M:6d} {C:4d} {HID:5d} {S1:3d} {S2:3d} | {t_ser:8.1f}       | {t_par:8.1f}                 | {t_sk:8.1f}', flush=True)
