"""Single-stream rANS decode / encode speed of the native host coder on a realistic symbol distribution (one latent block)."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import numpy as np, torch
from lvae import _native
from lvae.models.entropy_coding import DiscretizedGaussian
L = _native.lib()
dg = DiscretizedGaussian(cdf_form='erf'); dg.update()
q, ln, off = dg.host_tables()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 147456
g = np.random.default_rng(0)
idx = np.clip(g.normal(20, 8, n), 0, 63).astype(np.uint8)          # scale indexes around the middle of the table
scale = dg.scale_table.numpy()[idx]
sym = np.rint(g.normal(0, 1, n) * scale).astype(np.int32)
out = np.empty(8 * n + 64, np.uint8)
nb = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, n, q.ctypes.data, q.shape[1], ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
assert nb > 0
dec = np.empty(n, np.int32)
for name, fn in (('decode', lambda: L.lvae_rans_decode_with_indexes(out.ctypes.data, nb, idx.ctypes.data, n, q.ctypes.data, q.shape[1], ln.ctypes.data, off.ctypes.data, dec.ctypes.data)),
                 ('encode', lambda: L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, n, q.ctypes.data, q.shape[1], ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size))):
    fn()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        rc = fn()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name}: {n} symbols, {nb * 8 / n:.2f} bits/symbol, {dt * 1e3:.3f} ms, {dt / n * 1e9:.2f} ns/symbol')
assert np.array_equal(dec, sym)
