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


def explosive_line_source_halfspace(x, z, zs, times, alpha, beta, rho=1.0, area=1.0, wavelet=ricker, wavelet_dot=ricker_dot,
                                    period=3000.0, kmax_decades=14.0, t_window=None):
    """Particle velocity (v_x, v_z), z pointing DOWN from the free surface z = 0, at horizontal distance x and depth
    z >= 0 of an explosive line source at depth zs > 0 in a homogeneous half space ("Garvin's problem" with buried
    receivers) - the problem tests/explosive_source/explosive_source_lf4.py states, in the limit of a small source
    box and perfect absorbers.

    Direct wave: the closed form above.  Reflected P and converted SV waves (and with them the Rayleigh wave): plane-wave
    expansion with the free-surface conditions sigma_zz = sigma_xz = 0 solved per horizontal wavenumber k,

        phi  = S [ e^{-nu_a |z - zs|} / (2 nu_a)  +  A e^{-nu_a z} ],     psi = S B e^{-nu_b z},
        A = - e^{-nu_a zs} / (2 nu_a) * (W^2 + 4 k^2 nu_a nu_b) / D,       B = 4 i k nu_a W e^{-nu_a zs} / (2 nu_a D),
        W = 2 k^2 - w^2 / beta^2,    D = W^2 - 4 k^2 nu_a nu_b   (Rayleigh function),   nu_c = sqrt(k^2 - w^2 / c^2),

    displacement = grad(phi) + curl(psi e_y), summed over discrete wavenumbers k_n = 2 pi n / period (Bouchon's discrete
    wavenumber method: the images a distance `period` away arrive after the time window) at complex frequencies
    w + i gamma (which keeps the branch points and the Rayleigh pole off the path and damps the time wrap-around).
    Not part of the reference; an exact solution for the reference's own test problem."""
    times = np.asarray(times, dtype=np.float64)
    T = float(t_window) if t_window is not None else 2.0 * float(times.max()) + 1.0
    gamma = np.pi / T
    dw = 2.0 * np.pi / T
    # source spectrum at complex frequency: int r(t) e^{i w t} dt, r = wavelet (moment rate), by quadrature
    tq = np.linspace(0.0, min(T, 1.5), 6001)
    rq = wavelet(tq)
    nw = int(np.ceil(2.0 * np.pi * 25.0 / dw))                # up to 25 Hz: the Ricker of the test peaks at 4 Hz
    wr = dw * np.arange(nw + 1)
    w = wr + 1j * gamma
    trapz = getattr(np, "trapezoid", None) or np.trapz
    rhat = trapz(rq[None, :] * np.exp(1j * w[:, None] * tq[None, :]), tq, axis=1)
    S = area / (rho * alpha ** 2) * rhat                      # velocity = d/dt displacement: moment-rate spectrum
    dk = 2.0 * np.pi / period
    kmax = kmax_decades / max(z + zs, 0.25)                   # e^{-nu (z + zs)} < e^{-kmax_decades}
    n = int(np.ceil(kmax / dk))
    k = dk * np.arange(-n, n + 1)
    vx = np.zeros(len(w), dtype=complex)
    vz = np.zeros(len(w), dtype=complex)
    for m in range(len(w)):
        ka2, kb2 = (w[m] / alpha) ** 2, (w[m] / beta) ** 2
        na, nb = np.sqrt(k * k - ka2), np.sqrt(k * k - kb2)
        W = 2 * k * k - kb2
        D = W * W - 4 * k * k * na * nb
        inc0 = np.exp(-na * zs) / (2 * na)
        A = -inc0 * (W * W + 4 * k * k * na * nb) / D
        B = 4j * k * na * W * inc0 / D
        ea, eb = np.exp(-na * z), np.exp(-nb * z)
        wx = 1j * k * A * ea + nb * B * eb
        wz = -na * A * ea + 1j * k * B * eb
        ph = np.exp(1j * k * x)
        vx[m] = S[m] * dk / (2 * np.pi) * np.sum(wx * ph)
        vz[m] = S[m] * dk / (2 * np.pi) * np.sum(wz * ph)
    ex = np.exp(-1j * wr[:, None] * times[None, :])
    wgt = np.full(len(w), dw)
    wgt[0] = 0.5 * dw
    fx = np.exp(gamma * times) / np.pi * np.real(np.sum((wgt * vx)[:, None] * ex, axis=0))
    fz = np.exp(gamma * times) / np.pi * np.real(np.sum((wgt * vz)[:, None] * ex, axis=0))
    # direct wave (closed form); receiver at (x, z - zs) from the source
    r = float(np.hypot(x, z - zs))
    vr = explosive_line_source_2d(r, times, alpha, rho, area, wavelet_dot)
    return fx + vr * x / r, fz + vr * (z - zs) / r


def explosive_point_source_3d(r, times, alpha, rho=1.0, volume=1.0, wavelet=ricker, wavelet_dot=ricker_dot):
    """Radial particle velocity of an explosive point source in a homogeneous 3-D full space, formulation of
    seigen/elastic.py:204-219 with the source term  S_ij = volume * r(t) * delta^3(x) * delta_ij  (the limit of a small
    source box of that volume; BASELINE config 4's source): the displacement potential is
    phi = volume / (4 pi rho alpha^2 r) * R(t - r/alpha), R' = r, so

        v_r(r, t) = - volume / (4 pi rho alpha^2) * [ r(tau) / r^2 + r'(tau) / (alpha r) ],   tau = t - r / alpha."""
    times = np.asarray(times, dtype=np.float64)
    tau = times - r / alpha
    on = tau > 0
    out = np.zeros_like(times)
    out[on] = -volume / (4 * np.pi * rho * alpha ** 2) * (wavelet(tau[on]) / r ** 2 + wavelet_dot(tau[on]) / (alpha * r))
    return out


def explosive_box_source_halfspace(x, z, zs, times, alpha, beta, box=1.0, nq=3, **kw):
    """explosive_line_source_halfspace integrated over a square source box of edge `box` centred at (0, zs) - the exact
    field of the box source of explosive_source_lf4.py:36-40 with unit moment PER UNIT AREA times box^2 = `area`
    (Gauss-Legendre nq x nq over the source positions; the solution is linear in the source)."""
    gx, gw = np.polynomial.legendre.leggauss(nq)
    vx = np.zeros(len(times))
    vz = np.zeros(len(times))
    for a, wa in zip(gx, gw):
        for b, wb in zip(gx, gw):
            fx, fz = explosive_line_source_halfspace(x - 0.5 * box * a, z, zs + 0.5 * box * b, times, alpha, beta, **kw)
            vx += 0.25 * wa * wb * fx
            vz += 0.25 * wa * wb * fz
    return vx, vz
