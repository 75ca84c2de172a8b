#!/usr/bin/env python
"""rocprofv3 kernel trace vs the bench line of the SAME run: mean duration of every stage kernel over the
timed launches (the first `warmup` steps' launches of each kernel dropped) against the hipEvent averages
bench.py reports (roofline.kernels.*.avg_ms).

  python tools/compare_rocprof_bench.py <kernel_trace.csv> <bench line json> <warmup steps>
"""
import collections
import csv
import json
import sys


def main():
    trace, bench, warm = sys.argv[1], sys.argv[2], int(sys.argv[3])
    line = [l for l in open(bench) if l.startswith("{")][-1]
    b = json.loads(line)
    kern = b["roofline"]["kernels"]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"].replace("void ", "").replace("(sg::StageArgs)", "")
        dur[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    print("bench line: %.0f M DoF-updates/s, %d timed steps, %.3f ms/step" % (b["value"], b["steps"], b["ms_per_step"]))
    print("%-42s %8s %12s %12s %8s" % ("kernel", "launches", "rocprof ms", "hipEvent ms", "diff"))
    for name, v in kern.items():
        d = sorted(dur.get(name, []))
        per_step = v["launches"] // b["steps"]
        d = [x[1] for x in d[warm * per_step:]]          # drop the warm-up steps' launches
        if not d:
            print("%-42s not in the trace" % name)
            continue
        avg = sum(d) / len(d) / 1e6
        print("%-42s %8d %12.4f %12.4f %+7.1f%%" % (name, len(d), avg, v["avg_ms"], 100.0 * (v["avg_ms"] / avg - 1.0)))


if __name__ == "__main__":
    main()
