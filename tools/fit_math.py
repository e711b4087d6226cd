#!/usr/bin/env python3
"""Fit the polynomial coefficients used by the deterministic fp32 math
(pk_expf / pk_logf / pk_tanhf) shared -- as a written specification, not as
shared code -- by oracle/pk_oracle_math.h and parakeet.cpp_amd/csrc/pk_devmath.h.

Least squares on dense Chebyshev nodes in float64 (near-minimax); the achieved
fp32 accuracy is measured by tests/test_oracle_math.py against libm/float64.
Run:  python tools/fit_math.py   (prints C initialisers)
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(fn, a, b, deg, n=4000, weight=None):
    x = cheb_nodes(a, b, n)
    y = fn(x)
    V = np.vander(x, deg + 1, increasing=True)
    w = np.ones_like(x) if weight is None else weight(x)
    c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
    err = np.max(np.abs((V @ c - y) * w))
    return c, err

def show(name, c):
    print(f"// {name}")
    for i, v in enumerate(c):
        print(f"  c{i} = {np.float32(v)!r:>20}  /* {float(np.float32(v)).hex()} */")

# exp(r) = 1 + r + r^2 * E(r),  |r| <= ln2/2
L = np.log(2.0) / 2 * 1.0001
cE, e = fit(lambda r: np.where(r == 0, 0.5, (np.exp(r) - 1 - r) / np.where(r == 0, 1, r * r)), -L, L, 4)
show(f"exp: E(r) deg4, abs err {e:.3e}", cE)

# log(1+f) = f - f^2/2 + f^3 * Lg(f),  f in [sqrt(.5)-1, sqrt(2)-1]
a, b = np.sqrt(0.5) - 1 - 1e-4, np.sqrt(2.0) - 1 + 1e-4
def lg(f):
    f = np.where(np.abs(f) < 1e-9, 1e-9, f)
    return (np.log1p(f) - f + 0.5 * f * f) / (f ** 3)
cL, e = fit(lg, a, b, 8)
show(f"log: Lg(f) deg8, abs err {e:.3e}", cL)

# tanh(x) = x + x^3 * T(x^2),  |x| <= 0.55
def tg(z):
    x = np.sqrt(np.maximum(z, 1e-18))
    return (np.tanh(x) - x) / (x ** 3)
cT, e = fit(tg, 0.0, 0.55 ** 2 * 1.0001, 4)
show(f"tanh: T(z) deg4 (z=x^2), abs err {e:.3e}", cT)
