#!/usr/bin/env python
"""Config 5 (Marmousi 383 x 121, P3) seen from rocprofv3 --kernel-trace: where a step's time goes - inside the six stage
launches or between them - and how the items of a launch fall on the persistent grid.
usage: config5_timeline.py <kernel_trace.csv>   (trace of `python3 tools/bench_configs.py c5 --steps N`)"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tile2d_stage" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 6 * 6
rows = rows[len(rows) - min(n, 6 * 400):]          # the last 400 steps at most (steady state, graph replay)
# align to a step boundary: stage UH1 is tile2d_stage<P, 0, 0, ...> following a <P, 1, 1, ...> launch
def kind(r):
    a = r["Kernel_Name"].split("<")[1].split(",")
    return int(a[1]), int(a[2])
k0 = next(i for i in range(1, len(rows)) if kind(rows[i]) == (0, 0) and kind(rows[i - 1]) == (1, 1))
rows = rows[k0:]
rows = rows[:len(rows) // 6 * 6]
dur, gap = defaultdict(list), defaultdict(list)
names = ["UH1", "STEMP", "U1", "SH1", "UTEMP", "S1"]
for s in range(0, len(rows) - 6, 6):
    for j in range(6):
        r, nx = rows[s + j], rows[s + j + 1]
        dur[j].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        gap[j].append((int(nx["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
nsteps = len(dur[0])
step = (int(rows[6 * nsteps]["Start_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / nsteps
print("steps analysed: %d   step %.2f us (start of UH1 to start of the next UH1)" % (nsteps, step))
print("%-6s %-44s %10s %12s" % ("stage", "kernel", "inside us", "gap after us"))
tin = tgap = 0.0
for j in range(6):
    d, g = sum(dur[j]) / nsteps, sum(gap[j]) / nsteps
    tin += d
    tgap += g
    name = rows[j]["Kernel_Name"]
    print("%-6s %-44s %10.2f %12.2f" % (names[j], name[name.find("tile2d"):name.find(">") + 1], d, g))
print("inside the launches %.2f us = %.1f %% of the step; between them %.2f us = %.1f %%" % (tin, 100 * tin / step, tgap, 100 * tgap / step))
g0 = rows[0]
print("grid %s x workgroup %s" % (g0.get("Grid_Size_X", g0.get("Grid_Size", "?")), g0.get("Workgroup_Size_X", g0.get("Workgroup_Size", "?"))))
