#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_gpu.sh into profiles-ready files: kernel stats (CSV) and a JSON
with per-kernel average duration and HBM traffic per launch (PMC counters corrected by the calibration kernels)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "c2"


def short(name):
    n = re.sub(r"\(anonymous namespace\)::|dpck::", "", name)
    return re.sub(r"\(.*", "", n).replace("void ", "").strip()


def find(sub, pat):
    f = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return sorted(f)[-1] if f else None


def counter_avg(sub, counter):
    f = find(sub, "*counter_collection.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats = {}
f = find("trace", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_pct"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
            stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3)

# calibration: bytes actually moved / counter reading, per access width
true_bytes = float(1 << 30)
cal_f, cal_w = counter_avg("calib_fetch", "FETCH_SIZE"), counter_avg("calib_write", "WRITE_SIZE")
calib = {}
for k in ("calib_copy_dword", "calib_copy_dwordx4"):
    calib[k] = dict(fetch_counter=cal_f.get(k), write_counter=cal_w.get(k),
                    fetch_bytes_per_count=true_bytes / cal_f[k] if cal_f.get(k) else None,
                    write_bytes_per_count=true_bytes / cal_w[k] if cal_w.get(k) else None)

fetch, write = counter_avg("fetch", "FETCH_SIZE"), counter_avg("write", "WRITE_SIZE")
# which calibration applies: column kernels and locate use dword-per-lane streams, slab kernels 16 B per lane
width = {"k_zcol_fwd": "calib_copy_dword", "k_zcol_bwd": "calib_copy_dword", "k_zcol_fwdbwd": "calib_copy_dword", "k_locate": "calib_copy_dword",
         "k_splat_hw": "calib_copy_dwordx4", "k_splat_xl": "calib_copy_dwordx4", "k_gather_hw": "calib_copy_dwordx4", "k_loss_finalize": "calib_copy_dword"}
kern = {}
for k in sorted(set(fetch) | set(write)):
    base = re.sub(r"<.*", "", k)
    if not base.startswith("k_"):
        continue
    c = calib.get(width.get(base, "calib_copy_dword"), {})
    fb = fetch.get(k, 0.0) * (c.get("fetch_bytes_per_count") or 0.0)
    wb = write.get(k, 0.0) * (c.get("write_bytes_per_count") or 0.0)
    kern[k] = dict(avg_us=stats.get(k, {}).get("avg_us"), fetch_counter=fetch.get(k), write_counter=write.get(k),
                   hbm_read_bytes=fb, hbm_write_bytes=wb, hbm_bytes_per_launch=fb + wb)
json.dump(dict(config=config, calibration=calib, kernels=kern, kernel_stats=stats), open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in kern.items():
    print("%-28s avg %8s us   HBM read %8.2f MB  write %8.2f MB" % (k, "%.1f" % v["avg_us"] if v["avg_us"] else "?", v["hbm_read_bytes"] / 1e6, v["hbm_write_bytes"] / 1e6))
print("calibration", json.dumps(calib))
