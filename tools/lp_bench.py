"""gemm_lp_kernel timings on model shapes.  python tools/lp_bench.py [reps]; LP_ONLY=i selects one shape"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_mxfp8
L = _native.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(196608, 192, 384, 2), (196608, 384, 192, 1), (49152, 768, 384, 1), (49152, 384, 768, 2), (196608, 128, 192, 2), (12288, 1024, 512, 1)]
only = os.environ.get('LP_ONLY')
if only:
    shapes = [shapes[int(only)]]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, epi) in shapes:
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    Wt = torch.randn(N, K) / K ** 0.5
    wq = pack_mxfp8(Wt).cuda()
    bias, gamma = torch.randn(N, device='cuda'), torch.rand(N, device='cuda')
    res = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt16, d.ldw = A.data_ptr(), K, K, wq.data_ptr(), (K + 63) // 64 * 64
    d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec, d.a_bf16, d.out_bf16 = M, N, K, epi, 3, 1, 1
    for _ in range(3):
        assert L.lvae_gemm_f32(ctypes.byref(d), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lvae_gemm_f32(ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    by = 2 * M * K + N * K + 2 * M * N * (2 if epi in (2, 3) else 1)
    print(f'M={M} N={N} K={K} epi={epi}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s  {2.0 * M * N * K / us / 1e6:6.1f} TF/s', flush=True)
