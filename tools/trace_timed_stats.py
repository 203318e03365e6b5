#!/usr/bin/env python3
"""Per-kernel duration statistics of the TIMED launches of a profiled bench.py run.

`rocprofv3 --kernel-trace --stats` averages every dispatch of a kernel, including bench.py's untimed clock
spin-up and warm-up launches, which run at a lower clock.  This reads the per-dispatch
kernel trace of the same run and reports, per kernel, the statistics of the LAST `steps` dispatches only --
the ones inside bench.py's timed region -- next to the all-dispatch figures.

    python tools/trace_timed_stats.py <dir with *kernel_trace.csv> <steps> [dispatches per step] > timed_kernel_stats.csv
"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

root, steps = sys.argv[1], int(sys.argv[2])
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # dispatches of a kernel per bench step
paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = defaultdict(list)
for path in paths:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "AllAverageNs", "TimedCalls", "TimedAverageNs", "TimedMedianNs", "TimedMinNs",
            "TimedMaxNs"])
for name, d in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d.sort()
    dur = [e - s for s, e in d]
    # bench.py's timed steps are the last thing it launches on this kernel; kernels with fewer dispatches
    # than that (set-up) are reported whole
    timed = dur[-steps * per_step:] if len(dur) >= steps * per_step else dur
    w.writerow([name, len(dur), round(statistics.mean(dur), 1), len(timed), round(statistics.mean(timed), 1),
                round(statistics.median(timed), 1), min(timed), max(timed)])
