#!/bin/bash
# SQ counters of the step's kernels (separate rocprofv3 --pmc passes, kernel trace only): how busy the LDS pipe and the
# VALU are.  Runs on the GPU box via gpurun; output gpurun_out/pmc/summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CONFIG=${1:-c2}
OUT=gpurun_out/pmc_$CONFIG; rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --config $CONFIG --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-graph"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1; echo "pass $i ($grp) exit=$?"
done
python3 - $OUT <<'PY'
import csv, glob, collections, re, os
import sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|dpck::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "").strip()
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in sorted(agg.items()):
        line = k + "  " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items()))
        print(line); fh.write(line + "\n")
PY
