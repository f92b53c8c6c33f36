#!/usr/bin/env python3
"""Kernel durations AND the gaps between consecutive kernels of the step, from a rocprofv3 --kernel-trace CSV:
   python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("<")[0].split("::")[-1]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na, nb = short(a["Kernel_Name"]), short(b["Kernel_Name"])
    if not na.startswith("k_") or not nb.startswith("k_"):
        continue
    dur[na].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[na + " -> " + nb].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
for k, v in dur.items():
    print("%-40s n=%4d  median %6.2f us  mean %6.2f" % (k, len(v), med(v), sum(v) / len(v)))
for k, v in gap.items():
    print("gap %-36s n=%4d  median %6.2f us  mean %6.2f" % (k, len(v), med(v), sum(v) / len(v)))
tot = collections.defaultdict(list)
starts = [int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]) == "k_locate"]
d = [(b - a) / 1e3 for a, b in zip(starts, starts[1:])]
if d:
    print("k_locate start to next k_locate start: median %.2f us (n=%d)" % (med(d), len(d)))
