"""Benchmark record of an eigenmode run, as ``tests/eigenmode/eigenmode_bench.py`` of the
reference produces through pybench: the parameter series, the named timers and the DoF count /
error functionals as metadata.  pybench itself (file naming, plotting, cluster scripts) is not
reproduced; the record is one JSON object."""
import json

from seigen_amd.helpers import allreduce_sum
from seigen_amd.parallel import world
from seigen_amd.profiling import get_timers, timed_region
from seigen_amd.harness.eigenmode import Eigenmode2DLF4, Eigenmode3DLF4


def eigenmode_record(dim=3, N=3, degree=1, dt=0.125, T=2.0, solver='explicit', opt=2, path=None):
    """One run (eigenmode_bench.py:19-57).  Returns the record and writes it to `path` if given."""
    series = {'np': world()[1], 'dim': dim, 'size': N, 'T': T, 'solver': solver, 'opt': opt, 'degree': degree}
    # If dt is supressed (<0) infer it based on Courant number (eigenmode_bench.py:29-32)
    if dt < 0:
        dt = 0.5*(1.0/N)/(2.0**(degree-1))
    series['dt'] = dt
    get_timers(reset=True)
    if dim == 2:
        eigen = Eigenmode2DLF4(N, degree, dt, solver=solver, output=False)
        u1, s1 = eigen.eigenmode2d(T=T)
    elif dim == 3:
        eigen = Eigenmode3DLF4(N, degree, dt, solver=solver, output=False)
        u1, s1 = eigen.eigenmode3d(T=T)
    else:
        raise ValueError("dim must be 2 or 3")
    timings = {task: timer.total for task, timer in get_timers(reset=True).items()}
    meta = {'dofs': int(allreduce_sum(eigen.elastic.S.dof_count))}
    try:
        with timed_region('compute_error'):
            u_error, s_error = eigen.eigenmode_error(u1, s1)
        meta['u_error'], meta['s_error'] = float(u_error), float(s_error)
    except RuntimeError:
        meta['u_error'] = meta['s_error'] = 'NaN'
    timings['compute_error'] = get_timers(reset=True)['compute_error'].total
    record = {'benchmark': 'EigenmodeLF4', 'method': 'eigenmode', 'series': series, 'timings': timings, 'meta': meta}
    if path is not None and world()[0] == 0:
        with open(path, 'w') as f:
            json.dump(record, f, indent=1, sort_keys=True)
    return record


if __name__ == '__main__':
    print(json.dumps(eigenmode_record(N=4, degree=1, dt=0.125), indent=1, sort_keys=True))
