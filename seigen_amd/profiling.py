"""Named wall-clock regions, mirroring the ``pyop2.profiling.timed_region`` calls
of the reference (``seigen/elastic.py:76,121,247,278,286,291,299``) and the
``get_timers(reset=True)`` harvest of ``tests/eigenmode/eigenmode_bench.py:45-46``."""
import time
from contextlib import contextmanager

_timers = {}


class _Timer(object):
    def __init__(self):
        self.total = 0.0
        self.ncalls = 0


@contextmanager
def timed_region(name):
    t0 = time.perf_counter()
    try:
        yield
    finally:
        t = _timers.setdefault(name, _Timer())
        t.total += time.perf_counter() - t0
        t.ncalls += 1


def get_timers(reset=False):
    out = dict(_timers)
    if reset:
        _timers.clear()
    return out
