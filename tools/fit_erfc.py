"""Fit behind gelu_f / gelu_grad_f of csrc/common.h: erfc(t) = 2^-q(t) on [0, 4], q a degree-8 polynomial (weighted minimax fit by
iteratively re-weighted least squares), evaluated in emulated fp32 Horner form and checked against float64 on 4 M points."""
import numpy as np
from scipy.special import erfc

T, DEG, N = 4.0, 8, 6000
t = 0.5 * T * (1 - np.cos(np.pi * (np.arange(N) + 0.5) / N))
q, w = -np.log2(erfc(t)), erfc(t)
ww = w.copy()
for _ in range(200):
    c = np.polynomial.polynomial.polyfit(t, q, DEG, w=ww)
    e = w * (np.polynomial.polynomial.polyval(t, c) - q)
    ww = ww * (1 + 2 * np.abs(e) / np.abs(e).max()); ww /= ww.max()
c32 = c.astype(np.float32)
print("coefficients, low to high:", ", ".join("%.9ef" % v for v in c32))

def horner32(c, t):
    acc = np.full_like(t, c[-1], dtype=np.float32)
    for k in range(len(c) - 2, -1, -1):
        acc = (acc.astype(np.float64) * t.astype(np.float64) + np.float64(c[k])).astype(np.float32)  # fma: one rounding
    return acc

x = np.linspace(-12, 12, 4000001).astype(np.float32)
tv = np.minimum(np.abs(x) * np.float32(0.70710678), np.float32(T)).astype(np.float32)
e = np.exp2(-horner32(c32, tv).astype(np.float64)).astype(np.float32)
s = np.where(x >= 0, (np.float32(2) - e).astype(np.float32), e)
g = ((np.float32(0.5) * x).astype(np.float32) * s).astype(np.float32)
xd = x.astype(np.float64)
gref = 0.5 * xd * erfc(-xd / np.sqrt(2))
print("gelu: max |err| / max(1, |ref|) = %.3g" % (np.abs(g - gref) / np.maximum(1, np.abs(gref))).max())
pdf = (np.exp2(((x * x).astype(np.float32) * np.float32(-0.5 * 1.4426950408889634)).astype(np.float64)).astype(np.float32) * np.float32(0.3989422804)).astype(np.float32)
gg = ((np.float32(0.5) * s).astype(np.float64) + xd * pdf.astype(np.float64)).astype(np.float32)
ggref = 0.5 * erfc(-xd / np.sqrt(2)) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
print("gelu': max |err| = %.3g" % np.abs(gg - ggref).max())
