import numpy as np
from scipy.special import erf
K = 0.7978845608028654
def horner32(coef, w):
    acc = np.full_like(w, np.float32(coef[-1]), dtype=np.float32)
    for c in coef[-2::-1]:
        acc = (acc.astype(np.float64) * w + np.float32(c)).astype(np.float32)   # fused: one rounding
    return acc
def fit_odd(f, Z, deg_w, n=40001, iters=60):
    z = Z * np.cos(np.pi * (np.arange(n) + 0.5) / (2 * n))
    w = z * z
    A = np.stack([z * (w / Z**2) ** k for k in range(deg_w + 1)], 1)
    y = f(z)
    wt = np.ones_like(z)
    for it in range(iters):
        coef, *_ = np.linalg.lstsq(A * wt[:, None], y * wt, rcond=None)
        err = np.abs(A @ coef - y)
        wt = wt * (1 + 2 * err / err.max()); wt /= wt.max()
    coef = coef / np.array([Z ** (2 * k) for k in range(deg_w + 1)])   # polynomial in w = z^2 directly
    zz = np.linspace(0, Z, 800001)
    ww = (zz.astype(np.float32) * zz.astype(np.float32)).astype(np.float32)
    approx = zz.astype(np.float32) * horner32(coef, ww)
    return coef.astype(np.float32), np.abs(approx - f(zz)).max()
A_erf = lambda z: 0.5 * erf(z / np.sqrt(2))
B_erf = lambda z: 0.5 * erf(z / np.sqrt(2)) + z * np.exp(-z * z / 2) / np.sqrt(2 * np.pi)
def A_tanh(z): return 0.5 * np.tanh(K * z * (1 + 0.044715 * z * z))
def B_tanh(z):
    t = np.tanh(K * z * (1 + 0.044715 * z * z))
    return 0.5 * t + 0.5 * z * (1 - t * t) * K * (1 + 3 * 0.044715 * z * z)
for name, f, Z, degs in (("A_erf", A_erf, 4.0, (6, 7, 8)), ("B_erf", B_erf, 4.0, (7, 8, 9)), ("A_tanh", A_tanh, 3.5, (6, 7, 8, 9)), ("A_tanh", A_tanh, 4.0, (7, 8, 9)),
                         ("B_tanh", B_tanh, 3.5, (7, 8, 9, 10)), ("B_tanh", B_tanh, 4.0, (8, 9, 10))):
    for d in degs:
        coef, e32 = fit_odd(f, Z, d)
        print(name, "Z", Z, "deg", d, "err32 %.2e tail %.1e" % (e32, abs(abs(f(Z)) - 0.5)), "f(Z)=%.6f" % f(Z))
        print("   ", ", ".join("%.9ef" % c for c in coef))
