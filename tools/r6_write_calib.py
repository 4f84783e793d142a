"""Round 6: calibrate rocprofv3's WRITE_SIZE / FETCH_SIZE on KNOWN byte counts in this library's own store patterns (VERDICT r05 item 4;
MI355X_MICROARCH.md: 'WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own access pattern').  Launches, 20 times each:
  fill      -- torch fill_ of a 256 MiB fp32 buffer (plain 16-B streaming stores; nothing read)
  fc1       -- lvae_gemm_f32, f16x2, both operands pre-split, M = 98304, N = 384, K = 192, pre-split (H2K32) GELU output: whole-line 16-B stores
  fc2       -- the same GEMM family with the fp32 gamma / residual epilogue, N = 192, K = 384: 16-B stores after the quad transpose
  mlp_h2c   -- the fused MLP <192, 384> at M = 98304
  dwln      -- lvae_dwconv_ln_h2 on 4 x 128 x 192 x 192, k = 7 (pre-split output)
Run under `rocprofv3 --kernel-trace --pmc WRITE_SIZE` (and a second time with FETCH_SIZE); tools/r6_write_calib_summary.py prints
counter / known bytes per kernel."""
import ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_f16x2_k32
L = _native.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
known = {}
big = torch.empty(64 << 20, device='cuda')
for _ in range(20):
    big.fill_(1.0)
known['fill (vectorized_elementwise_kernel ... FillFunctor)'] = dict(write=big.numel() * 4, read=0)
M = 98304
g = torch.Generator().manual_seed(0)
def mk(M, N, K, epi, out_h2):
    A = torch.randn(M, K, generator=g).cuda(); Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda(); out = torch.empty(M, N, device='cuda')
    ah, wh = pack_f16x2_k32(A), pack_f16x2_k32(Wt)
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = ah.data_ptr(), K, K, Wt.data_ptr(), wh.data_ptr(), K
    d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec, d.a_h2, d.out_h2 = M, N, K, epi, 4, 1, out_h2
    d._keep = (A, Wt, bias, gamma, res, out, ah, wh)
    return d
d1 = mk(M, 384, 192, 1, 1)
for _ in range(20):
    assert L.lvae_gemm_f32(ctypes.byref(d1), st()) == 0
known['fc1 (gemm_h2p_kernel, pre-split GELU output)'] = dict(write=M * 384 * 4, read=M * 192 * 4 + 384 * 192 * 4)
d2 = mk(M, 192, 384, 2, 0)
for _ in range(20):
    assert L.lvae_gemm_f32(ctypes.byref(d2), st()) == 0
known['fc2 (gemm_h2p_kernel, fp32 output + residual)'] = dict(write=M * 192 * 4, read=M * 384 * 4 + 192 * 384 * 4 + M * 192 * 4)
C, HID = 192, 384
yf = torch.randn(M, C, generator=g).cuda(); W1 = (torch.randn(HID, C, generator=g) / C ** 0.5).cuda(); W2 = (torch.randn(C, HID, generator=g) / HID ** 0.5).cuda()
b1, b2, gamma = torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda(), torch.rand(C, generator=g).cuda()
res = torch.randn(M, C, generator=g).cuda(); out = torch.empty(M, C, device='cuda')
y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
dm = _native.MlpDesc()
dm.y, dm.w1, dm.b1, dm.w2, dm.b2, dm.gamma = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr()
dm.res, dm.out, dm.M, dm.C, dm.hid = res.data_ptr(), out.data_ptr(), M, C, HID
for _ in range(20):
    assert L.lvae_mlp_h2f(ctypes.byref(dm), st()) == 0
known['mlp_h2c_kernel<192, 384>'] = dict(write=M * C * 4, read=2 * M * C * 4 + 2 * C * HID * 4)
B, H, W, k = 4, 128, 192, 7
x = torch.randn(B, H, W, C, generator=g).cuda(); w = torch.randn(k * k, C, generator=g).cuda(); bb = torch.randn(C, generator=g).cuda()
sh, sc = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda(); yo = torch.empty_like(x)
for _ in range(20):
    assert L.lvae_dwconv_ln_h2(x.data_ptr(), w.data_ptr(), bb.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), yo.data_ptr(), B, H, W, C, k, st()) == 0
known['dwconv_ln_cl_kernel<7, ...> (pre-split output)'] = dict(write=B * H * W * C * 4, read=B * H * W * C * 4)
torch.cuda.synchronize()
json.dump(known, open(os.path.join(os.environ.get('CALIB_OUT', '/tmp'), 'write_calib_known.json'), 'w'), indent=1)
