"""Decoder A/B on saved streams (build box or GPU box host): python tools/rans_lab.py streams.npz lib1.so [lib2.so ...]
Each lib = a build of csrc/rans_host.cpp alone (clang++ -O3 -shared); single stream, single thread, best of 9."""
import ctypes, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
from lvae.models.entropy_coding import DiscretizedGaussian
dg = DiscretizedGaussian(); dg.update()
q, ln, off = dg.host_tables()
d = np.load(sys.argv[1])
libs = [(p, ctypes.CDLL(os.path.abspath(p))) for p in sys.argv[2:]]
for _, L in libs:
    L.lvae_rans_encode_with_indexes.restype = ctypes.c_long
    L.lvae_rans_encode_with_indexes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.lvae_rans_decode_with_indexes.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
nb = len([k for k in d.files if k.startswith('sym')])
tot = {p: 0.0 for p, _ in libs}
nsym = 0
for li in range(nb):
    sym, idx = np.ascontiguousarray(d[f'sym{li}']), np.ascontiguousarray(d[f'idx{li}'])
    n = sym.size; nsym += n
    out = np.empty(8 * n + 64, np.uint8)
    ref = None
    line = f'block {li} n={n:7d}'
    for p, L in libs:
        m = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, n, q.ctypes.data, q.shape[1], ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
        assert m > 0
        b = out[:m].tobytes()
        if ref is None: ref = b
        assert b == ref, 'encoders disagree'
        dec = np.empty(n, np.int32)
        best = 1e9
        for _ in range(9):
            t0 = time.perf_counter()
            rc = L.lvae_rans_decode_with_indexes(out.ctypes.data, m, idx.ctypes.data, n, q.ctypes.data, q.shape[1], ln.ctypes.data, off.ctypes.data, dec.ctypes.data)
            best = min(best, time.perf_counter() - t0)
        assert rc == 0 and np.array_equal(dec, sym), (p, rc)
        tot[p] += best
        line += f'  {best / n * 1e9:6.2f}'
    print(line + '  ns/symbol')
print('image total: ' + '  '.join(f'{p.split("/")[-1]} {t * 1e3:.3f} ms ({t / nsym * 1e9:.2f} ns/symbol)' for p, t in tot.items()))
