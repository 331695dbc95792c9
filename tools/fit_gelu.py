"""Derive / verify the polynomial used by gelu_erf_f2 (openvision_amd/csrc/common.h)."""
import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as P
X, deg = 4.0, 7
k = np.arange(deg + 1); t = np.cos(np.pi * (k + 0.5) / (deg + 1))
u = (t + 1) * X * X / 2; xs = np.sqrt(u)
coef = C.chebfit(t, (0.5 * (1 + erf(xs / np.sqrt(2))) - 0.5) / xs, deg)
qu = np.zeros(1)
for i, c in enumerate(C.cheb2poly(coef)):
    qu = P.polyadd(qu, c * P.polypow(np.array([-1.0, 2 / (X * X)]), i))
print("q(u) coefficients, u^0..u^7:", ", ".join("%.9e" % c for c in qu))
x = np.linspace(-16, 16, 640001).astype(np.float32)
xc = np.clip(x, -4, 4); uf = xc * xc
acc = np.full_like(uf, np.float32(qu[-1]))
for c in qu[-2::-1]:
    acc = acc * uf + np.float32(c)
g = x * (np.float32(0.5) + xc * acc)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
inside = np.abs(x) <= 4
print("max |GELU err| on [-4,4]: %.3e ; max |err|/|x| outside: %.3e" %
      (np.abs(g - ref)[inside].max(), (np.abs(g - ref)[~inside] / np.abs(x[~inside])).max()))
