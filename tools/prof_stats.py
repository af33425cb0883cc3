#!/usr/bin/env python3
"""Dev helper: print a rocprofv3 kernel_stats.csv (first match under a directory) as a table."""
import csv, glob, sys
import os
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:9.1f} max={float(r['MaxNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
