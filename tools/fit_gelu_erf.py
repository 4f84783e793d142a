"""Fit of the one-branch erf of csrc/device_math.h (lvae_gelu_erf1): erf(z) = sign(z) (1 - 2^(t q(t))), t = min(|z|, 4); prints the degree sweep,
the fp32-evaluated error of erf and of GELU, and the coefficients (Lawson-weighted least squares towards the minimax fit).  CPU only."""
import numpy as np
from scipy.special import erf, erfc
import numpy.polynomial.chebyshev as C
np.set_printoptions(precision=17)
LOG2E = 1.4426950408889634
def fit(N, T, iters=30):
    # find q (degree N-1) so that p(t) = t*q(t) ~ log2(erfc(t)) on [0,T], minimising max |erfc(t)*(2^(p - log2 erfc) - 1)| ~ erfc*ln2*|dp|
    t = np.linspace(0, T, 40001)[1:]
    target = np.log2(erfc(t)) / t           # q(t)
    w = erfc(t) * np.log(2) * t             # d(erf) = w * dq
    # iteratively reweighted least squares towards minimax (Lawson)
    lw = np.ones_like(t)
    x = 2 * t / T - 1
    V = C.chebvander(x, N - 1)
    best = None
    for it in range(iters):
        W = w * np.sqrt(lw)
        coef, *_ = np.linalg.lstsq(V * W[:, None], target * W, rcond=None)
        err = np.abs((V @ coef - target) * w)
        if best is None or err.max() < best[0]:
            best = (err.max(), coef.copy())
        lw = lw * (err / err.max() + 1e-3)
        lw /= lw.sum()
    return best
for T in (3.9, 4.0, 4.1):
    for N in (8, 9, 10, 11):
        e, c = fit(N, T)
        print(T, N, f'{e:.3e}')

print('---- fp32 evaluation')
def f32(x): return np.float32(x)
def fma32(a, b, c):   # a*b+c in fp32 with one rounding (emulated through fp64; double rounding is rare)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
def eval_erf(coef_mono, T, z):
    # z: float32 array (already x * 0.7071); erf(z) = copysign(1 - exp2(t * q(t)), z), t = min(|z|, T)
    t = np.minimum(np.abs(z), np.float32(T)).astype(np.float32)
    c = [np.float32(v) for v in coef_mono]      # q(t) = c[0] + c[1] t + ... (degree N-1)
    r = np.full_like(t, c[-1])
    for k in range(len(c) - 2, -1, -1):
        r = fma32(r, t, np.full_like(t, c[k]))
    p = (r.astype(np.float64) * t.astype(np.float64)).astype(np.float32)
    e = np.exp2(p.astype(np.float64)).astype(np.float32)     # v_exp_f32 ~ 1 ulp; idealised here
    return np.copysign((np.float32(1.0) - e).astype(np.float32), z)
T = 4.0
for N in (7, 8, 9, 10):
    e, cheb = fit(N, T)
    mono = np.polynomial.polynomial.Polynomial(C.cheb2poly(cheb))
    # cheb in x = 2t/T - 1 -> monomial in t
    x_of_t = np.polynomial.polynomial.Polynomial([-1.0, 2.0 / T])
    q = mono(x_of_t)
    coef = q.coef
    z = np.concatenate([np.linspace(0, 6, 2000001), np.logspace(-30, 0, 100001)]).astype(np.float32)
    got = eval_erf(coef, T, z)
    ref = erf(z.astype(np.float64))
    err = np.abs(got.astype(np.float64) - ref)
    i = err.argmax()
    print(N, f'fit {e:.2e}  fp32 max abs err {err.max():.3e} at z={z[i]:.5f}', 'coef', [float(np.float32(v)) for v in coef])

print('---- gelu N=8')
T = 4.0
e8, cheb = fit(8, T, iters=60)
mono = np.polynomial.polynomial.Polynomial(C.cheb2poly(cheb))
q = mono(np.polynomial.polynomial.Polynomial([-1.0, 2.0 / T]))
coef = [np.float32(v) for v in q.coef]
print('coef hex', [v.tobytes()[::-1].hex() for v in coef], [repr(float(v)) for v in coef])
x = np.concatenate([np.linspace(-8, 8, 4000001), np.linspace(-1.5, 1.5, 2000001)]).astype(np.float32)
z = (x * np.float32(0.70710678118654752440)).astype(np.float32)
er = eval_erf(coef, T, z)
g = ((np.float32(0.5) * x).astype(np.float32) * (np.float32(1.0) + er).astype(np.float32)).astype(np.float32)
xd = x.astype(np.float64)
ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
err = np.abs(g.astype(np.float64) - ref) / np.maximum(1.0, np.abs(xd))
print('gelu max err/max(1,|x|)', err.max(), 'at', x[err.argmax()])
