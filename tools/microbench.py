#!/usr/bin/env python
"""GPU micro-benchmarks of the native kernels on the shapes qarv_base uses (B images of 512x768).
    python tools/microbench.py [gemm|dw|all] [B]
Prints per-shape time, TFLOP/s (GEMM, algorithmic 2MNK) or GB/s (dwconv+LN, algorithmic read+write of the map)."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch  # noqa: E402
from lvae import _native  # noqa: E402
from lvae._native import GemmDesc  # noqa: E402

if os.environ.get('LVAE_LIB'):            # experimental build (tools/build_exp.sh)
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
L = _native.lib()


def st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm(M, N, K, epi):
    A = torch.randn(M, K, device='cuda')
    Wt = torch.randn(N, K, device='cuda') / K ** 0.5
    bias, gamma = torch.randn(N, device='cuda'), torch.rand(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    d = GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.ldw, d.bias, d.gamma = A.data_ptr(), K, K, Wt.data_ptr(), K, bias.data_ptr(), gamma.data_ptr()
    d.res, d.ldres, d.out, d.ldo, d.M, d.N, d.K, d.epi = res.data_ptr(), N, out.data_ptr(), N, M, N, K, epi
    prec = int(os.environ.get('LVAE_PREC', '0'))
    if prec:
        from lvae.models.base import pack_bf16x3, pack_f16x2
        w16 = pack_bf16x3(Wt) if prec == 2 else pack_f16x2(Wt) if prec == 4 else Wt.to(torch.bfloat16).contiguous()
        d.Wt16, d.prec = w16.data_ptr(), prec
        d._keep = w16
        if prec == 4 and os.environ.get('LVAE_H2P'):       # both operands pre-split (csrc/gemm_h2p.hip); LVAE_H2P = tile (42 41 22 21) or 1
            from lvae.models.base import pack_f16x2_k32
            ah, wh = pack_f16x2_k32(A), pack_f16x2_k32(Wt)
            d.A0, d.Wt16, d.a_h2 = ah.data_ptr(), wh.data_ptr(), 1
            d.cfg = int(os.environ['LVAE_H2P']) if int(os.environ['LVAE_H2P']) > 1 else 0
            d.out_h2 = int(os.environ.get('LVAE_OUT_H2', '0')) if epi in (0, 1) else 0
            d._keep = (ah, wh)
    if os.environ.get('LVAE_CONV3'):                       # "H,W": 3x3-tap gather over a (M / (H W), H, W, K / 9) map
        Hh, Ww = (int(v) for v in os.environ['LVAE_CONV3'].split(','))
        assert K % 9 == 0 and M % (Hh * Ww) == 0
        x = torch.randn(M, K // 9, device='cuda')
        d.A0, d.lda0, d.K0, d.a_mode, d.H, d.W = x.data_ptr(), K // 9, K // 9, 2, Hh, Ww
        d._keepx = x
    if os.environ.get('LVAE_CFG'):
        d.cfg = int(os.environ['LVAE_CFG'])
    if int(os.environ.get('LVAE_KSPLIT', '0')) > 1:
        S = int(os.environ['LVAE_KSPLIT'])
        ws = torch.empty(S * M * N, device='cuda')
        d.ksplit, d.ws = S, ws.data_ptr()
        d._keepws = ws
    t = timeit(lambda: L.lvae_gemm_f32(ctypes.byref(d), st()))
    return t


def bench_dw(B, H, W, C, k):
    x = torch.randn(B, H, W, C, device='cuda')
    w = torch.randn(k * k, C, device='cuda')
    b, sh, sc = torch.randn(C, device='cuda'), torch.randn(C, device='cuda'), torch.randn(C, device='cuda')
    y = torch.empty_like(x)
    t = timeit(lambda: L.lvae_dwconv_ln_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(),
                                            y.data_ptr(), B, H, W, C, k, st()))
    return t


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    px = {4: 24576, 8: 6144, 16: 1536, 32: 384, 64: 96}
    if what in ('gemm', 'all'):
        print(f'--- GEMM (fc1: epi=GELU, fc2: epi=gamma+res), B={B}')
        shapes = [(4, 192, 384), (8, 384, 768), (16, 512, 1024), (32, 512, 1024), (64, 512, 1024), (64, 512, 2048),
                  (32, 512, 1536), (16, 384, 768), (8, 256, 448), (8, 256, 512), (4, 128, 192)]
        tot_t = tot_f = 0
        for s, C, Hd in shapes:
            M = B * px[s]
            for (n, k, epi, nm) in ((Hd, C, 1, 'fc1'), (C, Hd, 2, 'fc2')):
                t = bench_gemm(M, n, k, epi)
                fl = 2.0 * M * n * k
                print(f's{s:<2} {nm} M={M:7d} N={n:5d} K={k:5d}  {t * 1e6:9.1f} us  {fl / t / 1e12:7.2f} TF/s')
                tot_t += t; tot_f += fl
        print(f'total {tot_t * 1e3:.2f} ms, {tot_f / tot_t / 1e12:.2f} TF/s aggregate')
    if what in ('dw', 'all'):
        print(f'--- dwconv+LN+AdaLN, B={B}')
        for (s, C, k) in [(4, 192, 7), (8, 384, 7), (16, 512, 5), (16, 512, 7), (32, 512, 3), (64, 512, 1), (16, 384, 5),
                          (8, 256, 7), (4, 128, 7), (8, 384, 7)]:
            H, W = 512 // s, 768 // s
            t = bench_dw(B, H, W, C, k)
            byts = 2.0 * B * H * W * C * 4
            print(f's{s:<2} C={C:4d} k={k}  {t * 1e6:9.1f} us  {byts / t / 1e9:8.1f} GB/s (algorithmic r+w)')




def gemmx():
    """Diagnostic sweep: main-loop efficiency vs K, epilogue cost."""
    for (M, N, K, epi) in [(49152, 768, 384, 0), (49152, 768, 384, 1), (49152, 768, 4096, 0), (49152, 768, 4096, 1),
                           (8192, 768, 4096, 0), (16384, 4096, 4096, 0), (49152, 384, 768, 0), (49152, 384, 768, 2),
                           (196608, 384, 192, 0), (196608, 384, 192, 1), (196608, 128, 192, 0), (196608, 128, 2048, 0)]:
        t = bench_gemm(M, N, K, epi)
        print(f'M={M:7d} N={N:5d} K={K:5d} epi={epi}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.2f} TF/s')


def gemmsk(B=8):
    """Small-map MLP layers (stride 32 / 64), batch B: parallel split-K on the fp32 operand (gemm_h2 + reduce launch) against the
    serial form on pre-split operands (gemm_h2p FOLD), slice count from the per-image rule (engine.auto_ksplit)."""
    from lvae.engine import auto_ksplit
    from lvae.models.base import pack_f16x2, pack_f16x2_k32
    for (rows, N, K, epi) in [(384, 1024, 512, 1), (384, 512, 1024, 2), (384, 1536, 512, 1), (384, 512, 1536, 2),
                              (96, 1024, 512, 1), (96, 512, 1024, 2), (96, 2048, 512, 1), (96, 512, 2048, 2)]:
        M = B * rows
        S = auto_ksplit(rows, N, K, 0, N, N, 4)
        A = torch.randn(M, K, device='cuda')
        Wt = torch.randn(N, K, device='cuda') / K ** 0.5
        bias, gamma, res = torch.randn(N, device='cuda'), torch.rand(N, device='cuda'), torch.randn(M, N, device='cuda')
        out, ws = torch.empty(M, N, device='cuda'), torch.empty(max(1, S) * M * N, device='cuda')
        keep = []

        def desc(a_h2):
            d = GemmDesc()
            w16 = pack_f16x2_k32(Wt) if a_h2 else pack_f16x2(Wt)
            a = pack_f16x2_k32(A) if a_h2 else A
            keep.extend([w16, a])
            d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw, d.bias, d.gamma = a.data_ptr(), K, K, Wt.data_ptr(), w16.data_ptr(), K, bias.data_ptr(), gamma.data_ptr()
            d.res, d.ldres, d.out, d.ldo, d.M, d.N, d.K, d.epi, d.prec, d.a_h2 = res.data_ptr(), N, out.data_ptr(), N, M, N, K, epi, 4, a_h2
            if S > 1:
                d.ksplit, d.ws = S, ws.data_ptr()
            return d
        dp, ds = desc(0), desc(1)
        tp = timeit(lambda: L.lvae_gemm_f32(ctypes.byref(dp), st()))
        ts = timeit(lambda: L.lvae_gemm_f32(ctypes.byref(ds), st()))
        print(f'rows/img={rows:4d} B={B} M={M:5d} N={N:5d} K={K:5d} epi={epi} S={S}: parallel {tp * 1e6:7.1f} us   serial {ts * 1e6:7.1f} us')


def mlpf():
    """C = 128 / hidden = 192 MLP (decoder stride-4 blocks): two pre-split GEMM launches against the fused kernel (csrc/mlp_h2c.hip)."""
    from lvae._native import MlpDesc
    from lvae.models.base import pack_f16x2_k32
    C, HID = (int(v) for v in os.environ.get('LVAE_MLP_SHAPE', '128,192').split(','))
    for M in (int(v) for v in os.environ.get('LVAE_MLP_MS', '196608,98304,49152,24576').split(',')):
        yf = torch.randn(M, C, device='cuda')
        W1, W2 = torch.randn(HID, C, device='cuda') / C ** 0.5, torch.randn(C, HID, device='cuda') / HID ** 0.5
        b1, b2, gamma = torch.randn(HID, device='cuda'), torch.randn(C, device='cuda'), torch.rand(C, device='cuda')
        res, out, hid = torch.randn(M, C, device='cuda'), torch.empty(M, C, device='cuda'), torch.empty(M, HID, device='cuda')
        y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
        d1, d2 = GemmDesc(), GemmDesc()
        d1.A0, d1.lda0, d1.K0, d1.Wt, d1.Wt16, d1.ldw, d1.bias, d1.out, d1.ldo = y.data_ptr(), C, C, W1.data_ptr(), w1h.data_ptr(), C, b1.data_ptr(), hid.data_ptr(), HID
        d1.M, d1.N, d1.K, d1.epi, d1.prec, d1.a_h2, d1.out_h2 = M, HID, C, 1, 4, 1, 1
        d2.A0, d2.lda0, d2.K0, d2.Wt, d2.Wt16, d2.ldw, d2.bias, d2.gamma = hid.data_ptr(), HID, HID, W2.data_ptr(), w2h.data_ptr(), HID, b2.data_ptr(), gamma.data_ptr()
        d2.res, d2.ldres, d2.out, d2.ldo, d2.M, d2.N, d2.K, d2.epi, d2.prec, d2.a_h2 = res.data_ptr(), C, out.data_ptr(), C, M, C, HID, 2, 4, 1
        m = MlpDesc()
        m.y, m.w1, m.b1, m.w2, m.b2, m.gamma, m.res, m.out = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr(), res.data_ptr(), out.data_ptr()
        m.M, m.C, m.hid = M, C, HID
        t2 = timeit(lambda: (L.lvae_gemm_f32(ctypes.byref(d1), st()), L.lvae_gemm_f32(ctypes.byref(d2), st())))
        two = out.clone()
        out.zero_()
        t1 = timeit(lambda: L.lvae_mlp_h2f(ctypes.byref(m), st()))
        torch.cuda.synchronize()
        print(f'        fused == two launches bit for bit: {bool((out.view(torch.int32) == two.view(torch.int32)).all())}')
        print(f'M={M:7d}: fc1 + fc2 launches {t2 * 1e6:7.1f} us   fused {t1 * 1e6:7.1f} us   ({4.0 * M * C * HID / t1 / 1e12:6.1f} TF/s, {12.0 * M * C / t1 / 1e12:5.2f} TB/s of y + res + out)')


def gemms4():
    """The stride-4 MLP layers (memory-bound, shallow K): per tile code via LVAE_H2P."""
    for (M, N, K, epi) in [(196608, 192, 128, 1), (196608, 128, 192, 2), (196608, 384, 192, 1), (196608, 192, 384, 2),
                           (98304, 192, 128, 1), (98304, 128, 192, 2), (24576, 192, 128, 1), (24576, 128, 192, 2)]:
        t = bench_gemm(M, N, K, epi)
        by = 4.0 * M * (K + N) + (4.0 * M * N if epi == 2 else 0)
        print(f'M={M:7d} N={N:5d} K={K:5d} epi={epi}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.2f} TF/s  {by / t / 1e12:5.2f} TB/s')


def mlppanel():
    """Does the hidden map of an MLP have to travel through HBM?  fc1 -> fc2 of a ConvNeXt block as two pre-split GEMM launches over the
    whole map, against the same launches over P row panels with ONE panel-sized hidden buffer that is written and re-read while it is
    still in the 256 MiB Infinity Cache (rows are independent: same bits).  Working set around it as in the model: y, res, out full size."""
    from lvae.models.base import pack_f16x2_k32
    for (C, HID, M) in [(192, 384, 196608), (192, 384, 98304), (384, 768, 49152), (384, 768, 24576), (256, 448, 49152), (128, 192, 196608)]:
        yf = torch.randn(M, C, device='cuda')
        W1, W2 = torch.randn(HID, C, device='cuda') / C ** 0.5, torch.randn(C, HID, device='cuda') / HID ** 0.5
        b1, b2, gamma = torch.randn(HID, device='cuda'), torch.randn(C, device='cuda'), torch.rand(C, device='cuda')
        res, out = torch.randn(M, C, device='cuda'), torch.empty(M, C, device='cuda')
        y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
        flush = torch.empty(96 * 1024 * 1024, device='cuda')             # 384 MB written between repetitions: nothing of the last run is cached
        row = []
        for P in (1, 2, 4, 8, 16):
            mp = M // P
            hid = torch.empty(mp, HID, device='cuda')
            descs = []
            for i in range(P):
                d1, d2 = GemmDesc(), GemmDesc()
                d1.A0, d1.lda0, d1.K0, d1.Wt, d1.Wt16, d1.ldw, d1.bias, d1.out, d1.ldo = y.data_ptr() + i * mp * C * 4, C, C, W1.data_ptr(), w1h.data_ptr(), C, b1.data_ptr(), hid.data_ptr(), HID
                d1.M, d1.N, d1.K, d1.epi, d1.prec, d1.a_h2, d1.out_h2 = mp, HID, C, 1, 4, 1, 1
                d2.A0, d2.lda0, d2.K0, d2.Wt, d2.Wt16, d2.ldw, d2.bias, d2.gamma = hid.data_ptr(), HID, HID, W2.data_ptr(), w2h.data_ptr(), HID, b2.data_ptr(), gamma.data_ptr()
                d2.res, d2.ldres, d2.out, d2.ldo = res.data_ptr() + i * mp * C * 4, C, out.data_ptr() + i * mp * C * 4, C
                d2.M, d2.N, d2.K, d2.epi, d2.prec, d2.a_h2 = mp, C, HID, 2, 4, 1
                descs.append((d1, d2))

            def run():
                flush.fill_(1.0)
                for d1, d2 in descs:
                    L.lvae_gemm_f32(ctypes.byref(d1), st()); L.lvae_gemm_f32(ctypes.byref(d2), st())
            t_all = timeit(run, iters=10)
            t_flush = timeit(lambda: flush.fill_(1.0), iters=10)
            row.append((P, (t_all - t_flush) * 1e6))
        base = row[0][1]
        print(f'C={C} hid={HID} M={M}: ' + '  '.join(f'P={p}: {t:7.1f} us ({base / t:4.2f}x)' for p, t in row) + f'   [{4.0 * M * C * HID / (min(t for _, t in row) * 1e-6) / 1e12:.0f} TF/s best]')


def mlptrace():
    """In-kernel timeline of csrc/mlp_h2c.hip (experimental build with -DH2C_EXP_TRACE, LVAE_LIB=...): wave 0 of workgroup 0, second tile."""
    from lvae._native import MlpDesc
    from lvae.models.base import pack_f16x2_k32
    C, HID = (int(v) for v in os.environ.get('LVAE_MLP_SHAPE', '192,384').split(','))
    M = int(os.environ.get('LVAE_TRACE_M', '196608'))
    HC = {(192, 384): 128, (128, 192): 64, (384, 768): 128}[(C, HID)]          # the instance's hidden chunk / column parts (csrc/mlp_h2c.hip)
    NSUB = 3 if C == 384 else 1
    yf = torch.randn(M, C, device='cuda')
    W1, W2 = torch.randn(HID, C, device='cuda') / C ** 0.5, torch.randn(C, HID, device='cuda') / HID ** 0.5
    b1, b2, gamma = torch.randn(HID, device='cuda'), torch.randn(C, device='cuda'), torch.rand(C, device='cuda')
    res = torch.randn(M, C, device='cuda')
    out = torch.zeros(M * C + 4096, device='cuda')
    y, w1h, w2h = pack_f16x2_k32(yf), pack_f16x2_k32(W1), pack_f16x2_k32(W2)
    m = MlpDesc()
    m.y, m.w1, m.b1, m.w2, m.b2, m.gamma, m.res, m.out = y.data_ptr(), w1h.data_ptr(), b1.data_ptr(), w2h.data_ptr(), b2.data_ptr(), gamma.data_ptr(), res.data_ptr(), out.data_ptr()
    m.M, m.C, m.hid = M, C, HID
    for _ in range(5):
        L.lvae_mlp_h2f(ctypes.byref(m), st())
    torch.cuda.synchronize()
    t = out[M * C:].view(torch.int64)[:128].cpu().numpy().astype('int64')
    t0 = t[0]
    KS1 = C // 32
    PT = KS1 + (HC // 32) * NSUB
    NP = (HID // HC) * PT
    assert 3 * NP <= 96, 'trace slots'
    print(f'mlp_h2c<{C}, {HID}, {HC}>: pos kind : wait_dma  barrier  stage(behind barrier -> next stage begins)   [cycles, s_memtime]')
    for P in range(NP):
        nxt = t[3 * (P + 1)] if P < NP - 1 else t[104]
        kind = ('F%d' % (P % PT)) if P % PT < KS1 else ('G%d' % (P % PT - KS1))
        extra = ''
        if P % PT == KS1 - 1:
            ch = P // PT
            extra = f'   incl. GELU phase {t[97 + 2 * ch] - t[96 + 2 * ch]}'
        print(f'{P:3d} {kind:3s} : {t[3 * P + 1] - t[3 * P]:8d} {t[3 * P + 2] - t[3 * P + 1]:8d} {nxt - t[3 * P + 2]:8d}{extra}')
    print(f'epilogue: compute {t[105] - t[104]} (incl. wait for the next tile\'s first stages), stores {t[106] - t[105]};  tile total {t[106] - t0} cycles')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'gemmx':
        gemmx()
    elif len(sys.argv) > 1 and sys.argv[1] == 'gemms4':
        gemms4()
    elif len(sys.argv) > 1 and sys.argv[1] == 'mlpf':
        mlpf()
    elif len(sys.argv) > 1 and sys.argv[1] == 'mlppanel':
        mlppanel()
    elif len(sys.argv) > 1 and sys.argv[1] == 'mlptrace':
        mlptrace()
    elif len(sys.argv) > 1 and sys.argv[1] == 'gemmsk_b':
        gemmsk(int(sys.argv[2]))
    elif len(sys.argv) > 1 and sys.argv[1] == 'gemmsk':
        for b in (1, 2, 4, 8, 16):
            gemmsk(b)
    elif len(sys.argv) > 1 and sys.argv[1] == 'gemm1':
        M, N, K, epi = [int(v) for v in sys.argv[2:6]]
        t = bench_gemm(M, N, K, epi)
        print(f'M={M} N={N} K={K} epi={epi} {t * 1e6:.1f} us {2.0 * M * N * K / t / 1e12:.2f} TF/s')
    else:
        main()
