import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
xs = np.concatenate([np.linspace(-5.2, 5.2, 2081)])
def gel(x): return x * 0.5 * (1 + erf(x / np.sqrt(2)))
gelu = gel(xs)
xd = np.linspace(-14, 14, 112001); geld = gel(xd)
p, q, cl = 3, 3, 5.0
def f(c, x, dt=np.float64):
    x = x.astype(dt); c = c.astype(dt)
    t = np.clip(x, -cl, cl); u = t * t
    P = ((c[3] * u + c[2]) * u + c[1]) * u + c[0]
    Q = ((c[6] * u + c[5]) * u + c[4]) * u + dt(1)
    return x * (dt(0.5) + t * P / Q)
c = np.array([0.3988795371, 0.02943036715, 0.003753279372, 2.996651815e-05, 0.2401170828, 0.02463599491, 0.001093515736])
wgt = np.ones_like(xs); best = (c, np.abs(f(c, xd) - geld).max())
for it in range(200):
    r = least_squares(lambda cc: (f(cc, xs) - gelu) * wgt, c, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=100)
    c = r.x
    e = np.abs(f(c, xd) - geld).max()
    if e < best[1]: best = (c.copy(), e)
    ee = np.abs(f(c, xs) - gelu); wgt = wgt * (1 + 1.5 * ee / ee.max()); wgt /= wgt.mean()
c, e = best
print('best max abs err (float64 eval) %.3e' % e); print([float('%.9g' % v) for v in c])
c32 = c.astype(np.float32)
e32 = np.abs(f(c32, xd, np.float32).astype(np.float64) - geld)
print('float32 evaluation: max abs err %.3e at x=%.3f; rel to max(|gelu|,1e-2): %.3e' % (e32.max(), xd[e32.argmax()], (e32 / np.maximum(np.abs(geld), 1e-2)).max()))
