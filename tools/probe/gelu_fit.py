"""Coefficients of the GELU / GELU-derivative forms of csrc/common.h (gelu_erf, gelu_erf2, gelu_erf_grad) and their error against
the exact functions in emulated fp32 arithmetic.

  h(x) = Phi(-x) = 2^-(1 + x R(x))          R of degree 5, discrete minimax (LP) of the error weighted by max(Phi(-x), 5e-4):
                                            absolute where h is large, relative in the tail (the output is 16-bit: u h must round like the exact value)
  G(x) = Phi(-x) - x phi(x) = 2^(-x^2 log2(e)/2) N(x),  N(0) = 1/2, N of degree 10, minimax (Lawson) of the absolute error

usage: python tools/probe/gelu_fit.py          (CPU, numpy + scipy)
"""
import numpy as np
from scipy.optimize import linprog
from scipy.special import erfc

f32 = np.float32
Phi = lambda u: 0.5 * erfc(-u / np.sqrt(2))
phi = lambda u: np.exp(-u * u / 2) / np.sqrt(2 * np.pi)
X = 6.0


def nodes(n):
    return np.sort(0.5 * X * (1 + np.cos(np.pi * (np.arange(n) + 0.5) / n)))


def minimax_lp(V, y, w):
    n, d = V.shape
    A = np.block([[V * w[:, None], -np.ones((n, 1))], [-V * w[:, None], -np.ones((n, 1))]])
    b = np.concatenate([y * w, -y * w])
    cost = np.zeros(d + 1)
    cost[-1] = 1
    r = linprog(cost, A_ub=A, b_ub=b, bounds=[(None, None)] * d + [(0, None)], method="highs")
    return r.x[:d], r.x[-1]


def lawson(V, y, w, iters=400):
    lw = np.ones(len(y))
    best = None
    for _ in range(iters):
        W = np.sqrt(lw) * w
        c, *_ = np.linalg.lstsq(V * W[:, None], y * W, rcond=None)
        e = np.abs((V @ c - y) * w)
        if best is None or e.max() < best[1]:
            best = (c.copy(), e.max())
        lw = lw * (e + 1e-300)
        lw /= lw.sum()
    return best


def fit_forward(deg=6, floor=1e-3):
    x = nodes(3000)
    T = -np.log2(Phi(-x))
    sc = X ** np.arange(1, deg + 1)
    V = np.stack([x ** k for k in range(1, deg + 1)], 1)
    c, _ = minimax_lp(V / sc, T - 1.0, np.log(2) * np.maximum(Phi(-x), floor * 0.5))
    return [float(f32(v)) for v in c / sc]


def fit_backward(deg=10):
    x = nodes(6000)
    N = (Phi(-x) - x * phi(x)) * np.exp(x * x / 2)
    V = np.stack([x ** k for k in range(1, deg + 1)], 1)
    c, _ = lawson(V, N - 0.5, np.exp(-x * x / 2))
    return [float(f32(v)) for v in c]


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def horner(cs, x, c0):
    p = np.full_like(x, f32(cs[-1]))
    for c in cs[-2::-1]:
        p = fma(p, x, np.full_like(x, f32(c)))
    return fma(p, x, np.full_like(x, f32(c0)))


def gelu_new(u, FW):
    x = np.minimum(np.abs(u), f32(6))
    h = np.exp2(-horner(FW, x, 1.0).astype(np.float64)).astype(f32)
    return fma(-x, h, np.maximum(u, f32(0)))


def dgelu_new(u, BW):
    x = np.minimum(np.abs(u), f32(6))
    e = np.exp2(((x * x).astype(f32).astype(np.float64) * f32(-0.72134752044448170368))).astype(f32)
    G = (e * horner(BW, x, 0.5)).astype(f32)
    return (f32(0.5) + np.copysign((f32(0.5) - G).astype(f32), u)).astype(f32)


def erf_as(x):      # the Abramowitz-Stegun 7.1.26 form used until round 4
    ax = np.abs(x)
    t = (1 / fma(np.full_like(ax, f32(0.3275911)), ax, np.ones_like(ax))).astype(f32)
    e = np.exp(-(ax * ax).astype(np.float64)).astype(f32)
    p = fma(np.full_like(t, f32(1.061405429)), t, np.full_like(t, f32(-1.453152027)))
    for c in (1.421413741, -0.284496736, 0.254829592):
        p = fma(p, t, np.full_like(t, f32(c)))
    return np.copysign((f32(1) - (p * t * e).astype(f32)).astype(f32), x), e


def bf16(x):
    b = x.astype(f32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7fff + ((b >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(f32)


def main():
    FW, BW = fit_forward(), fit_backward()
    print("forward  R coefficients x^1..x^6 :", FW)
    print("backward N coefficients x^1..x^10:", BW)
    u = np.linspace(-12, 12, 2400001).astype(f32)
    ud = u.astype(np.float64)
    g_ref, d_ref = ud * Phi(ud), Phi(ud) + ud * phi(ud)
    er, e = erf_as((u * f32(0.70710678118654752440)).astype(f32))
    g_old = (f32(0.5) * u * (f32(1) + er)).astype(f32)
    d_old = fma((u * f32(0.39894228040143267794)).astype(f32), e, (f32(0.5) * (f32(1) + er)).astype(f32))
    print("max |gelu error|  over [-12, 12]: new %.3e   Abramowitz-Stegun %.3e" % (np.abs(gelu_new(u, FW) - g_ref).max(), np.abs(g_old - g_ref).max()))
    print("max |gelu' error| over [-12, 12]: new %.3e   Abramowitz-Stegun %.3e" % (np.abs(dgelu_new(u, BW) - d_ref).max(), np.abs(d_old - d_ref).max()))
    ur = (np.random.default_rng(0).standard_normal(4000000) * 1.5).astype(f32)
    urd = ur.astype(np.float64)
    er, e = erf_as((ur * f32(0.70710678118654752440)).astype(f32))
    ref = bf16((urd * Phi(urd)).astype(f32))
    refd = bf16((Phi(urd) + urd * phi(urd)).astype(f32))
    print("bf16 results that differ from the rounded exact value, u ~ N(0, 1.5^2): gelu new %.3f %% / A-S %.3f %%;  gelu' new %.3f %% / A-S %.3f %%" % (
        100 * np.mean(bf16(gelu_new(ur, FW)) != ref), 100 * np.mean(bf16((f32(0.5) * ur * (f32(1) + er)).astype(f32)) != ref),
        100 * np.mean(bf16(dgelu_new(ur, BW)) != refd),
        100 * np.mean(bf16(fma((ur * f32(0.39894228040143267794)).astype(f32), e, (f32(0.5) * (f32(1) + er)).astype(f32))) != refd)))


if __name__ == "__main__":
    main()
