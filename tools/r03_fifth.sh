#!/bin/bash
# round 3: full-step rehearsals and rocprof profiles -> gpurun_out/r03/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines_fifth.jsonl; echo "$label done: $(echo $line | cut -c1-160)"; }
run "c4 full training step (BASELINE configs[3] as worded), one rank, eager" timeout -k 10 300 python bench.py --config c4 --full-step --steps 20 --warmup 3
run "c4 full training step, one rank, whole step as one HIP graph" timeout -k 10 300 python bench.py --config c4 --full-step --steps 20 --warmup 3 --captured
run "c4 full training step, two ranks over gloo on one GPU (functional rehearsal)" timeout -k 10 400 python bench.py --config c4 --full-step --gpus 2 --steps 4 --warmup 1 --rehearse-on-one-gpu
run "c3 full training step as one HIP graph, keep 0.07 (schedules followed step by step)" timeout -k 10 300 python bench.py --config c3 --steps 30 --warmup 5 --captured --keep 0.07
run "c3 full training step as one HIP graph, all points" timeout -k 10 300 python bench.py --config c3 --steps 30 --warmup 5 --captured
run "c5" timeout -k 10 300 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline
bash tools/profile_gpu.sh c2 > $OUT/profile_c2.log 2>&1; echo "profile c2 exit=$?"; tail -3 $OUT/profile_c2.log
bash tools/profile_gpu.sh c4 > $OUT/profile_c4.log 2>&1; echo "profile c4 exit=$?"; tail -3 $OUT/profile_c4.log
bash tools/profile_pmc.sh c2 > $OUT/pmc_c2.log 2>&1; echo "pmc exit=$?"; tail -6 $OUT/pmc_c2.log
