#!/usr/bin/env python3
"""Timeline of the last N microseconds of a rocprofv3 --kernel-trace --memory-copy-trace run (csv output):
    python3 tools/timeline.py <dir> [window_us]
Prints kernels and copies sorted by start time, relative to the first event in the window."""
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
if not ev:
    sys.exit("no events")
t_end = ev[-1][1]
sel = [e for e in ev if e[0] >= t_end - win * 1000]
t0 = sel[0][0]
for s, e, n, q in sel:
    print(f"{(s - t0) / 1000:9.1f} .. {(e - t0) / 1000:9.1f}  ({(e - s) / 1000:7.1f} us)  {n} {q}")
