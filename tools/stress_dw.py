"""Stress: the depthwise+LN kernel on one stream, split-K bf16x3 GEMMs on another; every output compared with a run alone."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_bf16x3
if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
L = _native.lib()
g = torch.Generator().manual_seed(3)
dw_cases = []
for (B, H, W, C, k) in [(1, 16, 24, 384, 7), (1, 8, 12, 512, 3), (1, 32, 48, 256, 7), (1, 16, 24, 512, 5), (1, 32, 48, 192, 7)]:
    x = torch.randn(B, H, W, C, generator=g).cuda(); wp = (torch.randn(k * k, C, generator=g) / k).cuda()
    b = torch.randn(C, generator=g).cuda(); sh = torch.randn(C, generator=g).cuda(); sc = (1 + 0.3 * torch.randn(C, generator=g)).cuda()
    dw_cases.append(dict(B=B, H=H, W=W, C=C, k=k, x=x, wp=wp, b=b, sh=sh, sc=sc))
def dw(c, y, st):
    assert L.lvae_dwconv_ln_f32(c['x'].data_ptr(), c['wp'].data_ptr(), c['b'].data_ptr(), None, None, c['sh'].data_ptr(), c['sc'].data_ptr(),
                                y.data_ptr(), c['B'], c['H'], c['W'], c['C'], c['k'], ctypes.c_void_p(st.cuda_stream)) == 0
gm_cases = []
for (M, N, K, S) in [(384, 512, 1024, 8), (96, 1024, 512, 4), (384, 1536, 512, 4), (1536, 384, 768, 2), (384, 512, 1536, 4)]:
    A = torch.randn(M, K, generator=g).cuda(); Wt = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    gm_cases.append(dict(M=M, N=N, K=K, S=S, A=A, Wt=Wt, W3=pack_bf16x3(Wt), bias=torch.randn(N, generator=g).cuda()))
ws = torch.empty(max(c['S'] * c['M'] * c['N'] for c in gm_cases), device='cuda'); cnt = torch.zeros(8192, dtype=torch.int32, device='cuda')
def gm(c, out, st, in_kernel=True):
    d = _native.GemmDesc()
    d.A0, d.lda0, d.K0, d.Wt, d.Wt16, d.ldw = c['A'].data_ptr(), c['K'], c['K'], c['Wt'].data_ptr(), c['W3'].data_ptr(), c['K']
    d.bias, d.out, d.ldo, d.M, d.N, d.K, d.epi, d.prec = c['bias'].data_ptr(), out.data_ptr(), c['N'], c['M'], c['N'], c['K'], 1, 2
    d.ksplit, d.ws, d.cnt = c['S'], ws.data_ptr(), (cnt.data_ptr() if in_kernel else None)
    assert L.lvae_gemm_f32(ctypes.byref(d), ctypes.c_void_p(st.cuda_stream)) == 0
cur = torch.cuda.current_stream()
dw_ref, gm_ref = [], []
for c in dw_cases:
    y = torch.empty_like(c['x']); dw(c, y, cur); torch.cuda.synchronize(); dw_ref.append(y)
for c in gm_cases:
    o = torch.empty(c['M'], c['N'], device='cuda'); gm(c, o, cur); torch.cuda.synchronize(); gm_ref.append(o)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else 'both'
bad_dw = bad_gm = n = 0
shown = 0
for rep in range(40):
    outs_dw, outs_gm = [], []
    for i in range(10):
        c = dw_cases[(i + rep) % len(dw_cases)]
        with torch.cuda.stream(s1):          # the fill must be ordered before the kernel
            y = torch.full_like(c['x'], float('nan'))
        dw(c, y, s1); outs_dw.append((y, dw_ref[(i + rep) % len(dw_cases)]))
        if mode == 'both':
            c2 = gm_cases[(i + 2 * rep) % len(gm_cases)]; o = torch.empty(c2['M'], c2['N'], device='cuda')
            gm(c2, o, s2); outs_gm.append((o, gm_ref[(i + 2 * rep) % len(gm_cases)]))
        elif mode == 'dwdw':
            c2 = dw_cases[(i + 2 * rep + 1) % len(dw_cases)]
            with torch.cuda.stream(s2):
                y2 = torch.full_like(c2['x'], float('nan'))
            dw(c2, y2, s2); outs_dw.append((y2, dw_ref[(i + 2 * rep + 1) % len(dw_cases)]))
    torch.cuda.synchronize()
    bad_dw += sum(0 if torch.equal(a, b) else 1 for a, b in outs_dw); bad_gm += sum(0 if torch.equal(a, b) else 1 for a, b in outs_gm)
    n += len(outs_dw)
    if shown < int(os.environ.get('STRESS_SHOW', '3')):
        for a, b in outs_dw:
            if not torch.equal(a, b) and shown < int(os.environ.get('STRESS_SHOW', '3')):
                shown += 1
                d = (a != b)
                px = d.any(dim=3).nonzero()
                print('  shape', tuple(a.shape), 'differing pixels (b,y,x):', px.tolist()[:8], 'n_px', len(px))
                bb, yy, xx = px[0].tolist()
                ch = d[bb, yy, xx].nonzero().flatten().tolist()
                print('    first pixel: differing channels', len(ch), 'of', a.shape[3], 'first', ch[:6], ' got', a[bb, yy, xx, ch[0]].item(), 'want', b[bb, yy, xx, ch[0]].item())
print(f'mode {mode}: dwconv outputs differing from the solo run: {bad_dw} of {n}; GEMM outputs differing: {bad_gm}')
if bad_dw:
    a, b = next((a, b) for a, b in outs_dw if not torch.equal(a, b)) if any(not torch.equal(a, b) for a, b in outs_dw) else (None, None)
    if a is not None:
        d = (a - b).abs(); print('last-rep example: max diff', float(d.max()), 'count', int((d > 0).sum()), 'nan', int(torch.isnan(a).sum()))
