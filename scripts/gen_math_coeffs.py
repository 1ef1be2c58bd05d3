"""Derives the polynomial coefficients used by include/rt_math.h (near-minimax fits at Chebyshev
nodes, computed in double precision with mpmath-free numpy, rounded to binary32) and prints them.
The header carries the printed values; tests/test_rt_math.py checks the accuracy they give."""
import numpy as np


def cheb_fit(f, lo, hi, degree, n=2000):
    k = np.arange(n)
    x = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / n)
    # least squares at Chebyshev nodes in a scaled variable for conditioning
    t = (x - 0.5 * (lo + hi)) / (0.5 * (hi - lo))
    c = np.polynomial.chebyshev.chebfit(t, f(x), degree)
    p = np.polynomial.chebyshev.cheb2poly(c)          # polynomial in t
    # substitute t = (x - mid) / half
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    poly = np.polynomial.Polynomial(p)(np.polynomial.Polynomial([-mid / half, 1.0 / half]))
    return poly.coef


def show(name, coef):
    print(name + " = { " + ", ".join("%.9ef" % float(np.float32(c)) for c in coef) + " }")


if __name__ == "__main__":
    # 2^f on [-0.5, 0.5]
    show("EXP2", cheb_fit(lambda x: np.exp2(x), -0.5, 0.5, 6))
    # e^r on [-0.35, 0.35]
    show("EXP", cheb_fit(lambda x: np.exp(x), -0.35, 0.35, 6))
    # log(m) = 2 s P(z), s = (m-1)/(m+1), z = s^2 in [0, 0.0295]; P(z) = atanh(s)/s
    def P(z):
        s = np.sqrt(np.maximum(z, 1e-300))
        return np.where(z < 1e-12, 1.0 + z / 3.0, np.arctanh(s) / s)
    show("LOG", cheb_fit(P, 0.0, 0.0296, 4))
    # asin(x) = x + x z R(z), z = x^2 in [0, 0.25]
    def R(z):
        x = np.sqrt(np.maximum(z, 1e-300))
        return np.where(z < 1e-10, 1.0 / 6.0 + 3.0 * z / 40.0, (np.arcsin(x) / x - 1.0) / np.maximum(z, 1e-300))
    show("ASIN", cheb_fit(R, 0.0, 0.25, 5))
