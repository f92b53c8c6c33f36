#!/bin/bash
# SQ counters of the step's kernels (separate rocprofv3 --pmc passes, kernel trace only): how busy the LDS pipe and the
# VALU are.  Runs on the GPU box via gpurun:  bash tools/profile_pmc.sh <config> [<library variant under scratch/> ...]
# ("main" or nothing = the in-tree library); output gpurun_out/pmc_<config>[_<variant>]/summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CONFIG=${1:-c2}; shift
[ $# -eq 0 ] && set -- main
for variant in "$@"; do
  OUT=gpurun_out/pmc_$CONFIG; [ "$variant" != main ] && OUT=${OUT}_$variant
  rm -rf $OUT; mkdir -p $OUT
  if [ "$variant" != main ]; then export DPC_RENDER_LIB="$PWD/scratch/$variant/libdpc_render.so"; else unset DPC_RENDER_LIB; fi
  BENCH="python3 bench.py --config $CONFIG --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-graph $PMC_BENCH_ARGS"   # e.g. PMC_BENCH_ARGS="--sigma-rel 3.0"
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1; echo "$variant pass $i ($grp) exit=$?"
  done
  python3 - $OUT $variant <<'PY'
import csv, glob, collections, re, os
import sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|dpck::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "").strip()
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|dpck::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k).replace("void ", "").strip()
        if k.startswith("k_"):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
with open(out + "/summary.txt", "w") as fh:
    fh.write("# library variant: %s (kernel us: under the counter pass, for orientation only)\n" % sys.argv[2])
    for k, cs in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        extra = ""
        if m.get("SQ_LDS_IDX_ACTIVE"):
            extra += "  LDS_conflict_share=%.3f" % (m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"])
        if m.get("SQ_WAVE_CYCLES"):
            extra += "  wait_share=%.3f" % (m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"])
        if dur.get(k):
            extra += "  us_under_pmc=%.2f" % (sorted(dur[k])[len(dur[k]) // 2])
        line = k + "  " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())) + extra
        print(line); fh.write(line + "\n")
PY
done
