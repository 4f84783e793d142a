"""Bit-equality of the persistent loader-wave GEMM study (tools/studies/gemm_h2q.hip, cfg = 61) against gemm_h2p's 128 x 64 tile (cfg = 21).
Needs an experimental library that links the study in:
    EXTRA_SRC=gemm_h2q.hip tools/build_exp.sh h2q gemm_h2p.hip -DLVAE_EXP_H2Q;  LVAE_LIB=_bin/h2q/liblvae_hip.so python tools/h2q_equal.py
One tile, fewer tiles than workgroups, ragged M and N, two to eighteen tiles per persistent workgroup (the ring running across tile
boundaries, stores of one tile draining under the next), K = 64 ... 768, every epilogue, the pre-split output of fc1; three launches each."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lossy-vae_amd'))
from lvae import _native  # noqa: E402

if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
from lvae.models.base import pack_f16x2_k32  # noqa: E402


def gemm(ah, K, Wt, wh, bias, gamma, res, out, N, M, epi, oh2, cfg):
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = ah.data_ptr(), K, K, Wt.data_ptr(), wh.data_ptr(), K
    d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec, d.a_h2, d.out_h2, d.cfg = M, N, K, epi, 4, 1, oh2, cfg
    rc = _native.lib().lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0, rc


for (M, N, K, epi) in [(128, 64, 64, 0), (1000, 448, 256, 1), (300 * 128 + 5, 768, 384, 1), (49152, 768, 384, 1), (49152, 384, 768, 2),
                       (24576, 448, 256, 1), (131, 96, 384, 3), (98304, 192, 128, 1), (70001, 256, 448, 2), (12288, 1024, 512, 0)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.rand(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    ah, wh = pack_f16x2_k32(A), pack_f16x2_k32(Wt)
    oh2 = 1 if (epi == 1 and N % 32 == 0) else 0
    ref = torch.full((M, N), float('nan'), device='cuda')
    gemm(ah, K, Wt, wh, bias, gamma, res, ref, N, M, epi, oh2, 21)
    bad = 0
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device='cuda')
        gemm(ah, K, Wt, wh, bias, gamma, res, out, N, M, epi, oh2, 61)
        bad += int((out.view(torch.int32) != ref.view(torch.int32)).sum())
    print(f'M={M} N={N} K={K} epi={epi}: {bad} words differ from gemm_h2p over 3 launches')
