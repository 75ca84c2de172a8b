#!/usr/bin/env python
"""profiles/<round>/config3_traffic.json from the passes of tools/profile_config3.sh.

Bytes per launch of each stage kernel = FETCH_SIZE*1024/f + WRITE_SIZE*1024/w, where f and w are
the fractions of the true bytes the two counters report for this code's access pattern, measured
with tools/calib_fetch.hip in the same session (f = 1/2 on gfx950, as MI355X_MICROARCH.md says).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_summary import summarise  # noqa: E402
import glob


def counters(d, name):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for kernel, vals, n in summarise(f):
            if name in vals:
                out[kernel] = vals[name]
    return out


def main():
    root = sys.argv[1]
    fetch, write = counters(root + "/fetch", "FETCH_SIZE"), counters(root + "/write", "WRITE_SIZE")
    cf, cw = counters(root + "/calib_f", "FETCH_SIZE"), counters(root + "/calib_w", "WRITE_SIZE")
    moved_kb = 4.0 * 1024 * 1024   # calib_fetch moves 4 GiB per kernel
    f = w = None
    for k, v in cf.items():
        if k.startswith("read_kernel"):
            f = v / moved_kb
    for k, v in cw.items():
        if k.startswith("write_kernel"):
            w = v / moved_kb
    if f is None or w is None:
        f, w, how = 0.5, 1.0, "calibration pass missing: guide factors used"
    else:
        how = "calibrated in this session with tools/calib_fetch.hip"
    kern = {}
    for k in fetch:
        if "stage" not in k:
            continue
        rb, wb = fetch[k] * 1024 / f, write.get(k, 0.0) * 1024 / w
        kern[k] = {"read_bytes": rb, "write_bytes": wb, "bytes": rb + wb}
    print(json.dumps({
        "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 "
                "--warmup 1` (config 3, 1 GPU, tools/profile_config3.sh); counters are in KB; " + how +
                ": FETCH_SIZE reads %.4f of the true bytes, WRITE_SIZE %.4f; mean per launch." % (f, w),
        # which kernels these numbers belong to: bench.py quotes them only while the sources are unchanged
        "csrc_sha256": __import__("bench").csrc_digest(),
        "fetch_factor": f, "write_factor": w, "kernels": kern}, indent=1))


if __name__ == "__main__":
    main()
