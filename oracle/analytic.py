"""Closed-form wave fields used to pin the oracle's normalisation.  ORACLE (test infrastructure).

explosive_line_source_2d: particle velocity of an isotropic ("explosive") line source in a homogeneous 2-D
full space, in the formulation of seigen/elastic.py:204-219,

    rho v_t = div s,      s_t = lambda (div v) I + mu (grad v + grad v^T) + A r(t) delta(x) I,

i.e. the source term of explosive_source_lf4.py:36-40 in the limit of a small box of area A.  With the
displacement w = int v dt = grad(phi):  phi_tt - alpha^2 lap(phi) = (A / rho) R(t) delta(x), R' = r, and the 2-D
Green's function H(t - r/alpha) / (2 pi alpha^2 sqrt(t^2 - r^2/alpha^2)); substituting t - tau = (r/alpha) cosh(theta)
removes the square-root singularity:

    v_r(r, t) = - A / (2 pi rho alpha^3) * int_0^inf  r'(t - (r/alpha) cosh(theta)) cosh(theta) d theta.

Only P waves (an isotropic source in a full space radiates no shear).  Not part of the reference; an independent
pin of the source normalisation and of the P-wave propagation of the oracle / HIP path.
"""
import numpy as np


def ricker(t, a=159.42, t0=0.3):
    """(-1 + 2 a (t-t0)^2) exp(-a (t-t0)^2), explosive_source_lf4.py:37-38."""
    x = t - t0
    return (-1.0 + 2 * a * x * x) * np.exp(-a * x * x)


def ricker_dot(t, a=159.42, t0=0.3):
    x = t - t0
    return (6 * a * x - 4 * a * a * x ** 3) * np.exp(-a * x * x)


def explosive_line_source_2d(r, times, alpha, rho=1.0, area=1.0, wavelet_dot=ricker_dot, nq=2000):
    """Radial velocity v_r at distance r for the times given (wavelet taken as zero before t = 0)."""
    times = np.asarray(times, dtype=np.float64)
    out = np.zeros_like(times)
    xg, wg = np.polynomial.legendre.leggauss(nq)
    for k, t in enumerate(times):
        if t <= r / alpha:
            continue
        thmax = np.arccosh(alpha * t / r)
        th = 0.5 * thmax * (xg + 1.0)
        w = 0.5 * thmax * wg
        ch = np.cosh(th)
        out[k] = -area / (2 * np.pi * rho * alpha ** 3) * np.sum(w * wavelet_dot(t - (r / alpha) * ch) * ch)
    return out
