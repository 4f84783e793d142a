"""Bit-equality of the epilogue-interleaved persistent fc1 GEMM study (tools/studies/gemm_h2e.hip, cfg = 51) against gemm_h2p.
Needs an experimental library that links the study in:
    EXTRA_SRC=gemm_h2e.hip tools/build_exp.sh h2e gemm_h2p.hip -DLVAE_EXP_H2E;  LVAE_LIB=_bin/h2e/liblvae_hip.so python tools/h2e_equal.py
One tile, ragged M and N (448 = 7 column tiles, 96 = 1.5), fewer tiles than workgroups, and the model's launches (several tiles per
persistent workgroup; K = 256 / 384 / 512 = 8 / 12 / 16 stages).  Repeated: nothing may depend on what a previous launch left in LDS."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lossy-vae_amd'))
from lvae import _native  # noqa: E402

if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
from lvae.models.base import pack_f16x2_k32  # noqa: E402


def gemm(ah, K, Wt, wh, bias, out, N, M, cfg):
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = ah.data_ptr(), K, K, Wt.data_ptr(), wh.data_ptr(), K
    d.bias, d.out, d.ldo = bias.data_ptr(), out.data_ptr(), N
    d.M, d.N, d.K, d.epi, d.prec, d.a_h2, d.out_h2, d.cfg = M, N, K, 1, 4, 1, 1, cfg
    rc = _native.lib().lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0, rc


for (M, N, K) in [(128, 64, 256), (1000, 448, 256), (300 * 128 + 5, 768, 384), (49152, 768, 384), (12288, 1024, 512), (24576, 512, 256), (131, 96, 384)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ah, wh = pack_f16x2_k32(A), pack_f16x2_k32(Wt)
    ref = torch.full((M, N), float('nan'), device='cuda')
    gemm(ah, K, Wt, wh, bias, ref, N, M, 0)
    bad = 0
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device='cuda')
        gemm(ah, K, Wt, wh, bias, out, N, M, 51)
        bad += int((out.view(torch.int32) != ref.view(torch.int32)).sum())
    print(f'M={M} N={N} K={K}: {bad} words differ from gemm_h2p over 3 launches')
