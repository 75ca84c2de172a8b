"""cProfile of ElasticLF4.run() on the reference's explosive-source set-up (120 x 60, P2, 2500 steps): where the host
side of a run goes next to the device's stepping."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.getcwd())
import seigen_amd, seigen_amd.helpers as helpers
import seigen_amd.harness.explosive_source as hx
helpers.log = seigen_amd.elastic.log = hx.log = lambda s: None
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4
es = ExplosiveSourceLF4()
el = es.setup(300.0, 150.0, 2.5, degree=2, dt=0.001)
el.run(0.01)
el.block.sync()
es = ExplosiveSourceLF4()
el = es.setup(300.0, 150.0, 2.5, degree=2, dt=0.001)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
el.run(2.5)
el.block.sync()
pr.disable()
print("run: %.4f s" % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
