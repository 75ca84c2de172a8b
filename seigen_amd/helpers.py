"""Helpers with the reference's names and semantics (``seigen/helpers.py``)."""
from math import sqrt

from .parallel import world, _dist


def log(s):
    """Rank-0 logging (``seigen/helpers.py:6-12``)."""
    if world()[0] == 0:
        print(s)


def Vp(mu, l, density):
    r"""P-wave velocity :math:`\sqrt{(\lambda + 2\mu)/\rho}` (``seigen/helpers.py:15-28``)."""
    return sqrt((l + 2 * mu) / density)


def Vs(mu, density):
    r"""S-wave velocity :math:`\sqrt{\mu/\rho}` (``seigen/helpers.py:31-43``)."""
    return sqrt(mu / density)


def cfl_dt(dx, Vp, courant_number):
    r"""Largest timestep allowed by the CFL condition (``seigen/helpers.py:46-54``)."""
    return (courant_number * dx) / Vp


def allreduce_sum(value):
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return value
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return type(value)(t.item())


def get_dofs(mesh, p):
    """Global (stress, velocity) degree-of-freedom counts (``seigen/helpers.py:57-67``;
    the reference body uses undefined names - this is what it is meant to return)."""
    from .functionspace import TensorFunctionSpace, VectorFunctionSpace
    S = TensorFunctionSpace(mesh, 'DG', p, name='S')
    U = VectorFunctionSpace(mesh, 'DG', p, name='U')
    return allreduce_sum(S.dof_count), allreduce_sum(U.dof_count)
