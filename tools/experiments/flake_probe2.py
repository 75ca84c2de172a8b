"""tests/test_harness_gpu.py::_multiblock_case (sponge + source: extras) repeated, with and without SEIGEN_HIP_GQ / TEAM."""
import os, sys
sys.path.insert(0, os.getcwd())
from tests.test_harness_gpu import _multiblock_case
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for env in ({"SEIGEN_HIP_GQ": "1"}, {}, {"SEIGEN_HIP_TEAM": "4"}):
    for k in ("SEIGEN_HIP_GQ", "SEIGEN_HIP_TEAM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for pipelined in (True, False):
        bad = 0
        for t in range(trials):
            try:
                _multiblock_case(3, 3, (9, 9, 9), (3, 3, 3), pipelined, extras=True)
            except AssertionError as e:
                bad += 1
                print(env, "pipelined", pipelined, "trial", t, "FAILED:", str(e)[:60], flush=True)
        print(env, "pipelined", pipelined, ":", bad, "failures in", trials, flush=True)
