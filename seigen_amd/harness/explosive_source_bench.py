"""Benchmark record of an explosive-source run, as ``tests/explosive_source/explosive_source_bench.py`` of the
reference produces through pybench: the parameter series (``h``, ``T``, ``explicit``, ``:16-19``) and the named
timers of the run (``:23-24``).  pybench itself is not reproduced; the record is one JSON object, like
``seigen_amd.harness.eigenmode_bench``."""
import json

from seigen_amd.helpers import allreduce_sum
from seigen_amd.parallel import world
from seigen_amd.profiling import get_timers
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4


def explosive_source_record(T=0.01, h=2.5, explicit=True, path=None, **setup_kw):
    """One run (explosive_source_bench.py:15-24).  `explicit` selects the solver string as the reference's flag does
    (True: 'explicit', False: 'implicit'); further keywords go to ``ExplosiveSourceLF4.setup`` (e.g. ``dt``, since
    the CFL step of explosive_source_lf4.py:30-32 is unstable with the explicit sponge).  Returns the record and
    writes it to `path` if given."""
    series = {'np': world()[1], 'h': h, 'T': T, 'explicit': bool(explicit)}
    get_timers(reset=True)
    es = ExplosiveSourceLF4()
    es.explosive_source_lf4(T=T, h=h, solver='explicit' if explicit else 'implicit', output=False, **setup_kw)
    timings = {task: timer.total for task, timer in get_timers(reset=True).items()}
    meta = {'dofs': int(allreduce_sum(es.elastic.S.dof_count)), 'steps': int(es.elastic.block.counters()['steps'])}
    record = {'benchmark': 'ExplosiveSourceLF4', 'method': 'explosive_source', 'series': series, 'timings': timings,
              'meta': meta}
    if path is not None and world()[0] == 0:
        with open(path, 'w') as f:
            json.dump(record, f, indent=1, sort_keys=True)
    return record


if __name__ == '__main__':
    print(json.dumps(explosive_source_record(dt=0.001), indent=1, sort_keys=True))
