"""Derive / verify the polynomial used by gelu_erf_grad_f2 (openvision_amd/csrc/common.h): d/dx gelu_erf(x) = Phi(x) + x phi(x)
= 0.5 + x_c s(x_c^2) on x_c = clamp(x, -X, X) (the odd part divided by x is even and smooth)."""
import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as P
for X, deg in ((4.0, 7), (4.5, 8), (5.0, 9), (5.0, 8), (4.5, 9)):
    k = np.arange(deg + 1); t = np.cos(np.pi * (k + 0.5) / (deg + 1))
    u = (t + 1) * X * X / 2; xs = np.sqrt(u)
    f = lambda x: 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
    coef = C.chebfit(t, (f(xs) - 0.5) / xs, deg)
    su = np.zeros(1)
    for i, c in enumerate(C.cheb2poly(coef)):
        su = P.polyadd(su, c * P.polypow(np.array([-1.0, 2 / (X * X)]), i))
    x = np.linspace(-16, 16, 640001).astype(np.float32)
    xc = np.clip(x, -X, X); uf = xc * xc
    acc = np.full_like(uf, np.float32(su[-1]))
    for c in su[-2::-1]:
        acc = acc * uf + np.float32(c)
    g = np.float32(0.5) + xc * acc
    ref = f(x.astype(np.float64))
    print("X=%.1f deg=%d: max |gelu' err| %.3e" % (X, deg, np.abs(g - ref).max()))
    print("   s(u) coefficients, u^0..: " + ", ".join("%.9e" % c for c in su))
