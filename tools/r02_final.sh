#!/bin/bash
# round-2 closing measurements on one GPU box: the full GPU test run, then every bench line quoted in DESIGN.md
# (one JSON line each, prefixed by a label) -> gpurun_out/r02_final/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_final; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "tests exit=$?"
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines.jsonl; echo "$label done"; }
run "driver-style default (20 steps, 5 warm-up), with cpu_baseline" python bench.py --steps 20 --warmup 5
run "driver-style again" python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run "c2 steady state (200 steps)" python bench.py --steps 200 --warmup 50 --no-cpu-baseline
run "c2 through the reference's plain call sequence" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --api plain
run "c4" python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline
run "c5" python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline
run "c3 full training step, all points" python bench.py --config c3 --steps 30 --warmup 5
run "c3 full training step, keep 0.07" python bench.py --config c3 --steps 30 --warmup 5 --keep 0.07
run "c3 full training step as one HIP graph, all points" python bench.py --config c3 --steps 30 --warmup 5 --captured
run "c3 full training step as one HIP graph, keep 0.07" python bench.py --config c3 --steps 30 --warmup 5 --captured --keep 0.07
run "c3 two ranks over gloo on one GPU (functional rehearsal)" python bench.py --config c3 --gpus 2 --steps 10 --warmup 2 --rehearse-on-one-gpu
python tools/bench_nearest.py > $OUT/bench_nearest.jsonl 2>>$OUT/bench.err; echo "nearest exit=$?"
bash tools/profile_c3.sh > $OUT/profile_c3.log 2>&1; echo "c3 profile exit=$?"; cp gpurun_out/prof_c3/kernel_stats_per_step.txt $OUT/ 2>/dev/null
