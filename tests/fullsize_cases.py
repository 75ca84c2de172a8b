"""BASELINE configs 2, 3 and 5 at FULL size as parity cases: the definitions shared by the oracle-side generator
(tests/golden/make_golden_fullsize.py -> tests/golden/fullsize_c{2,3,5}.npz) and the GPU tests
(tests/test_fullsize_oracle_gpu.py).  Only inputs live here (sizes, constants, closed-form initial states,
source wavelet, which cells are sampled); each side builds its own mesh, operators and step from them.

  c3  3-D eigenmode, 64^3 cubes x 6 tets, P4, FP64 (tests/eigenmode/eigenmode_3d.py:7-40), 3 LF4 steps
  c2  2-D explosive source, 512 x 512 squares, P2, DG4 sponge + box source (explosive_source_lf4.py:17-45), 20 steps
  c5  Marmousi 383 x 121 squares, P3, per-cell lambda / mu from seigen/marmousi.py's lookup and a per-cell Gardner
      density in the physical update, box source, 20 steps
c2 and c5 start from a smooth non-zero state (so that the sponge and every cell's material act from the first
step) and use the reference's Ricker wavelet centred ten steps into the run (so that the source is at full
strength inside the 20 steps)."""
import hashlib
import math

import numpy as np

A_RICKER = 159.42


def ricker(t, t0):
    return (-1.0 + 2 * A_RICKER * (t - t0) ** 2) * math.exp(-A_RICKER * (t - t0) ** 2)


def smooth_state(X, k, s_scale):
    """u_i = sin(k_i . x), s_ij = s_ji = s_scale cos(k_(i+j)%d . x + i - j) at node coordinates X [.., d]."""
    d = X.shape[-1]
    u = np.stack([np.sin(X @ k[i]) for i in range(d)], axis=-1)
    s = np.zeros(X.shape[:-1] + (d, d))
    for i in range(d):
        for j in range(i, d):
            s[..., i, j] = s[..., j, i] = s_scale * np.cos(X @ k[(i + j) % d] + i - j)
    return u, s


C3 = dict(n=64, P=4, steps=3, rho=1.0, mu=0.25, lam=0.5, dt=0.5 * (1.0 / 64) / 2 ** 3)   # eigenmode_3d.py:17-20, :80
_VP2 = math.sqrt((3599.3664 + 2 * 3600.0) / 1.0)
C2 = dict(n=512, h=2.5, P=2, steps=20, rho=1.0, mu=3600.0, lam=3599.3664, dt=0.05 * 2.5 / _VP2,
          sponge=20.0, sigma=1000.0, sigma_degree=4, src_half=0.5, src=(45.0, 512 * 2.5 - 1.0),
          k=np.array([[0.031, 0.047], [0.053, 0.022]]), s_scale=3600.0)
C5 = dict(P=3, steps=20, courant=0.05, src_half=12.0, k=np.array([[0.0020, 0.0030], [0.0035, 0.0015]]), s_scale=1.0e7)


def sample_cells(ncells, forced=(), count=400, seed=2024):
    rng = np.random.default_rng(seed)
    pick = rng.choice(ncells, size=min(count, ncells), replace=False)
    edge = [0, 1, ncells // 2, ncells - 2, ncells - 1]
    return np.unique(np.concatenate([pick, np.asarray(edge, dtype=np.int64), np.asarray(forced, dtype=np.int64)]))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def layer_sums(field, nlayers):
    """[nlayers, nd * ncomp]: sums over the cells of each slab of the slowest mesh axis (cells are numbered with
    that axis slowest on both sides) - a check of the whole field, not only of the sampled cells."""
    nc = field.shape[0]
    per = nc // nlayers
    return field.reshape(nlayers, per, -1).sum(axis=1)
