"""The polynomial of csrc/uce_epilogue.h gelu_erf: P(u) ~ log2(0.5 erfcx(u / sqrt 2)) on [0, 6.25] (Chebyshev least squares on Chebyshev
nodes, converted to the power basis), and the error of the resulting GELU in f32 arithmetic against fp64.   python tools/fit_gelu.py [degree]"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf, erfcx

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 7
UM = 6.25
xs = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * UM / 2 + UM / 2
coef = np.array(C.Chebyshev.fit(xs, np.log2(0.5 * erfcx(xs / np.sqrt(2))), deg, domain=[0, UM]).convert(kind=np.polynomial.Polynomial).coef)
g = np.concatenate([np.linspace(-12, 12, 600001), np.random.default_rng(0).normal(0, 2, 400000)]).astype(np.float32)
u = np.minimum(np.abs(g), np.float32(UM))
p = np.full_like(u, np.float32(coef[-1]))
for k in range(len(coef) - 2, -1, -1):
    p = (p * u + np.float32(coef[k])).astype(np.float32)
h = np.exp2(((u * u).astype(np.float32) * np.float32(-0.5 * 1.4426950408889634) + p).astype(np.float32)).astype(np.float32)
gel = (np.maximum(g, np.float32(0)) - u * h).astype(np.float32)
ref = g.astype(np.float64) * 0.5 * (1 + erf(g.astype(np.float64) / np.sqrt(2)))
err = np.abs(gel - ref)
print("coefficients (constant first):", ", ".join("%.9ef" % c for c in coef))
print("max abs %.3g   max rel (|g| < 6.25) %.3g" % (err.max(), (err / np.maximum(np.abs(ref), 1e-300))[np.abs(g) < UM].max()))
