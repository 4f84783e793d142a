"""Diagnostics for gemm_lp_kernel on the GPU box: simple exactly-representable cases."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_mxfp8, unpack_mxfp8
L = _native.lib()

def run(A, Wt, a_bf16=1, out_bf16=0):
    M, K = A.shape; N = Wt.shape[0]
    Ad = (A.to(torch.bfloat16) if a_bf16 else A.float()).cuda().contiguous()
    out = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16 if out_bf16 else torch.float32)
    wq = pack_mxfp8(Wt).cuda()
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt16, d.ldw = Ad.data_ptr(), K, K, wq.data_ptr(), (K + 63) // 64 * 64
    d.out, d.ldo, d.M, d.N, d.K, d.prec, d.a_bf16, d.out_bf16 = out.data_ptr(), N, M, N, K, 3, a_bf16, out_bf16
    rc = L.lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0, rc
    return out.float().cpu()

M, N, K = 128, 64, 64
o = run(torch.ones(M, K), torch.ones(N, K))
print('ones x ones (expect 64):', o.unique().tolist()[:8])
o = run(torch.ones(M, K) * 3, torch.ones(N, K) * 0.5)
print('3 x 0.5 (expect 96):', o.unique().tolist()[:8])
# k consistency: A one-hot at k0, W = identity
bad = []
for k0 in range(64):
    A = torch.zeros(M, K); A[:, k0] = 1
    o = run(A, torch.eye(64))
    exp = torch.zeros(M, N); exp[:, k0] = 1
    if not torch.equal(o, exp):
        bad.append((k0, o[0].nonzero().reshape(-1).tolist(), o[0][o[0] != 0].tolist()))
print('one-hot/identity mismatches:', len(bad), bad[:6])
# row mapping: A[i][:] = i+1 (exact in bf16 up to 128; fp8 quantises) with W = ones/64
A = (torch.arange(M).float() % 8 + 1).unsqueeze(1).expand(M, K).contiguous()
o = run(A, torch.ones(N, K))
print('row values (expect 64*(i%8+1)):', (o[:, 0] / 64).tolist()[:16])
# scales: row i scaled by 2^(i%5)
A = torch.pow(2.0, (torch.arange(M) % 5).float()).unsqueeze(1) * torch.ones(M, K)
o = run(A, torch.ones(N, K))
print('row scales (expect 64*2^(i%5)):', (o[:, 0] / 64).tolist()[:10])
# two halves with different magnitudes
A = torch.ones(M, K); A[:, 32:] = 16.0
Wt = torch.ones(N, K); Wt[:, 32:] = 0.0
print('only first half counted (expect 32):', run(A, Wt).unique().tolist()[:5])
Wt = torch.ones(N, K); Wt[:, :32] = 0.0
print('only second half counted (expect 512):', run(A, Wt).unique().tolist()[:5])
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).float(); Wt = torch.randn(N, K, generator=g)
ref = unpack_mxfp8(pack_mxfp8(A), M, K).double() @ unpack_mxfp8(pack_mxfp8(Wt), N, K).double().t()
o = run(A, Wt)
print('random max rel err:', float(((o.double() - ref).abs() / (ref.abs() + 1)).max()))
