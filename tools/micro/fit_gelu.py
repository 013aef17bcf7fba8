"""Fit of the GELU/erfc polynomial used by vs_gelu (csrc/vs_common.h): weighted minimax (Lawson) fit of -ln erfc(t) = t*Q(t) on [0,4], float32 validation."""
import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
T = 4.0
N = 20001
t = np.linspace(0, T, N)
# Chebyshev-like clustering
t = T * 0.5 * (1 - np.cos(np.linspace(0, np.pi, N)))
target = -np.log(erfc(t))          # P(t)
def fit(deg, iters=60):
    # P(t) = t*Q(t), Q degree deg-1
    V = np.stack([t ** (k + 1) for k in range(deg)], 1)
    w = erfc(t)                     # error in erf ~ erfc * dP
    lw = np.ones_like(t)
    for it in range(iters):
        W = w * lw
        c, *_ = np.linalg.lstsq(V * W[:, None], target * W, rcond=None)
        err = np.abs(np.exp(-(V @ c)) - erfc(t))
        lw = lw * (err / err.max() + 1e-3) ** 0.5      # Lawson-ish
        lw /= lw.max()
    return c, err.max()
for deg in range(7, 13):
    c, e = fit(deg)
    # float32 evaluation check (Horner, separate mul/add roundings ~ pessimistic vs fma)
    c32 = c.astype(np.float32)
    tt = np.linspace(0, 6, 600001).astype(np.float32)
    tc = np.minimum(tt, np.float32(T))
    acc = np.full_like(tc, c32[-1])
    for k in range(deg - 2, -1, -1):
        acc = (acc * tc + c32[k]).astype(np.float32)
    P = (acc * tc).astype(np.float32)
    approx = (np.float32(1) - np.exp2((-P * np.float32(1.4426950408889634)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    e32 = np.abs(approx.astype(np.float64) - erf(tt.astype(np.float64))).max()
    print(deg, "fit max abs err %.3e" % e, "float32 eval max abs err %.3e" % e32)
    if deg in (9, 10, 11):
        print("  coeffs:", ", ".join("%.9e" % x for x in c))

# ---- final folded form for GELU(v) = max(v,0) - 0.5*|v|*e,  e = exp2(u*Q(u)),  u = min(|v|, UMAX), coefficients of -log2e*P(u/sqrt2)
c, e = fit(9)
s2 = np.sqrt(2.0)
q = np.array([-1.4426950408889634 * c[k] / s2 ** (k + 1) for k in range(9)])
print("UMAX", 4.0 * s2)
print("q =", ", ".join("%.9ef" % x for x in q.astype(np.float32)))
q32 = q.astype(np.float32)
v = np.linspace(-9, 9, 1800001).astype(np.float32)
a = np.abs(v)
u = np.minimum(a, np.float32(4.0 * s2))
acc = np.full_like(u, q32[-1])
for k in range(7, -1, -1):
    acc = (acc * u + q32[k]).astype(np.float32)
ee = np.exp2((acc * u).astype(np.float32)).astype(np.float32)
g = (np.maximum(v, 0) - (np.float32(0.5) * a * ee).astype(np.float32)).astype(np.float32)
from scipy.special import erf as erf64
gref = 0.5 * v.astype(np.float64) * (1 + erf64(v.astype(np.float64) / s2))
err = np.abs(g.astype(np.float64) - gref)
print("GELU max abs err %.3e at v=%.4f; max rel-to-1ulp(|ref|,1e-30): %.2f" % (err.max(), v[err.argmax()], (err / np.maximum(np.abs(gref) * 1.19e-7, 1e-30)).max()))
import torch
gt = torch.nn.functional.gelu(torch.from_numpy(v)).numpy()
print("torch fp32 gelu vs fp64 ref max abs err %.3e" % np.abs(gt.astype(np.float64) - gref).max())
print("ours vs torch fp32 max abs diff %.3e" % np.abs(g - gt).max())
