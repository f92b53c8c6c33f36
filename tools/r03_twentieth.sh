#!/bin/bash
# what the 20-step window of the driver's call looks like on the GPU time line: kernel trace of bench.py --steps 20 --warmup 5
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03p; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1; echo "exit=$?"
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("<")[0].split("::")[-1]
loc = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_locate"]
print("k_locate launches:", len(loc))
t = [int(rows[i]["Start_Timestamp"]) for i in loc]
for k in range(len(loc)):
    end = int(rows[loc[k] + 3]["End_Timestamp"]) if loc[k] + 3 < len(rows) else 0
    names = [short(rows[loc[k] + j]["Kernel_Name"]) for j in range(4) if loc[k] + j < len(rows)]
    durs = [(int(rows[loc[k] + j]["End_Timestamp"]) - int(rows[loc[k] + j]["Start_Timestamp"])) / 1e3 for j in range(4) if loc[k] + j < len(rows)]
    gap = (t[k] - int(rows[loc[k] - 1]["End_Timestamp"])) / 1e3 if loc[k] > 0 else 0.0
    print("step %2d: gap before %9.2f us  step %6.2f us  kernels %s" % (k, gap, (end - t[k]) / 1e3, " ".join("%.1f" % d for d in durs)), names if k == 0 else "")
PY
tail -1 $OUT/run.log | cut -c1-200
