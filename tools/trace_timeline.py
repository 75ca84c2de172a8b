#!/usr/bin/env python
"""Timeline of one LF4 step from a rocprofv3 --kernel-trace CSV: start / end of every kernel dispatch relative to the
first one of the step, per queue - shows whether RCCL's send/receive kernel runs beside the SECOND launch of a split
stage or only after it.  usage: trace_timeline.py kernel_trace.csv [first_dispatch_index] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[skip]["Start_Timestamp"])
for r in rows[skip:skip + cnt]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"]
    name = name[:name.find("(")] if "(" in name else name
    print("%9.1f -> %9.1f us  (%7.1f)  q%-3s  %s" % (a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), name[:90]))
