#!/bin/bash
# Round-4 profile session (one gpurun call): rocprofv3 stats + HBM traffic passes for c2 and c4, SQ counters for c2, the
# kernel statistics of the reference's plain call sequence replayed as a graph.  Everything lands in gpurun_out/r04_profile/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_profile; rm -rf $OUT; mkdir -p $OUT
bash tools/profile_gpu.sh c2 > $OUT/profile_c2.log 2>&1; echo "profile c2 exit=$?"; tail -6 $OUT/profile_c2.log
cp gpurun_out/prof_c2/summary.json $OUT/rocprof_summary_c2.json; cp gpurun_out/prof_c2/kernel_stats.csv $OUT/kernel_stats_c2.csv
bash tools/profile_gpu.sh c4 > $OUT/profile_c4.log 2>&1; echo "profile c4 exit=$?"; tail -6 $OUT/profile_c4.log
cp gpurun_out/prof_c4/summary.json $OUT/rocprof_summary_c4.json; cp gpurun_out/prof_c4/kernel_stats.csv $OUT/kernel_stats_c4.csv
bash tools/profile_pmc.sh c2 > $OUT/pmc_c2.log 2>&1; echo "pmc exit=$?"; cp gpurun_out/pmc_c2/summary.txt $OUT/sq_counters_c2.txt; cat $OUT/sq_counters_c2.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plain -- python3 bench.py --api plain --launch graph --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/plain.log 2>&1; echo "plain trace exit=$?"
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = sorted(glob.glob(out + "/plain/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1]))) if f else []
with open(out + "/kernel_stats_plain_sequence.txt", "w") as fh:
    fh.write("# bench.py --api plain --launch graph --steps 100: pointcloud_project_fast + torch loss + backward, captured once, replayed\n")
    fh.write("# calls  avg_us  total_pct  kernel\n")
    for r in rows:
        fh.write("%6s %8.2f %6s  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:110]))
print(open(out + "/kernel_stats_plain_sequence.txt").read()[:3000])
PY
rm -rf $OUT/plain
