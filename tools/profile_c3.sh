#!/bin/bash
# rocprofv3 kernel statistics of the full training step (BASELINE configs[2]) as one HIP graph -> gpurun_out/prof_c3/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_c3; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config c3 --steps 30 --warmup 5 --captured "$@" > $OUT/trace.log 2>&1; echo "trace exit=$?"
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats_top.txt", "w") as fh:
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        line = "%6.2f%%  calls %6s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:150])
        print(line); fh.write(line + "\n")
    fh.write("total kernel time %.3f ms over the run\n" % (tot / 1e6))
PY
tail -2 $OUT/trace.log | cut -c1-300
