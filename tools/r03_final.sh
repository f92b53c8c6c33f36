#!/bin/bash
# round-3 closing measurements on one GPU box: the full GPU test run, then every bench line quoted in DESIGN.md
# (one JSON line each, prefixed by a label) -> gpurun_out/r03_final/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_final; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 $OUT/gpu_tests.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?"; tail -2 $OUT/smoke.log
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines.jsonl; echo "$label done"; }
run "driver-style default (20 steps, 5 warm-up), with cpu_baseline" timeout -k 10 300 python bench.py --steps 20 --warmup 5
run "driver-style again" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run "c2 steady state (200 steps)" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline
run "c2 steady state (200 steps), the same kernels as a replayed HIP graph (direct launch)" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --launch graph
run "c2 through the reference's plain call sequence" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --api plain --launch graph
run "c4" timeout -k 10 300 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline
run "c5" timeout -k 10 300 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline
run "c2 two ranks over gloo on one GPU (functional rehearsal of --gpus N)" timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu --no-cpu-baseline
run "c3 full training step, all points, eager" timeout -k 10 300 python bench.py --config c3 --steps 30 --warmup 5
run "c3 full training step as one HIP graph, all points" timeout -k 10 300 python bench.py --config c3 --steps 30 --warmup 5 --captured
run "c3 full training step as one HIP graph, keep 0.07, schedules followed" timeout -k 10 300 python bench.py --config c3 --steps 30 --warmup 5 --captured --keep 0.07
run "c3 two ranks over gloo on one GPU (functional rehearsal)" timeout -k 10 300 python bench.py --config c3 --gpus 2 --steps 10 --warmup 2 --rehearse-on-one-gpu
run "c4 full training step (BASELINE configs[3] as worded), one rank, eager" timeout -k 10 300 python bench.py --config c4 --full-step --steps 20 --warmup 3
run "c4 full training step, one rank, one HIP graph" timeout -k 10 300 python bench.py --config c4 --full-step --steps 20 --warmup 3 --captured
run "c4 full training step, two ranks over gloo on one GPU (functional rehearsal)" timeout -k 10 400 python bench.py --config c4 --full-step --gpus 2 --steps 4 --warmup 1 --rehearse-on-one-gpu
python - <<'PY'
import json
for l in open("gpurun_out/r03_final/bench_lines.jsonl"):
    d=json.loads(l); j=d['line']
    print("%-90s %10.0f  %.4f ms" % (d['label'][:90], j['value'], j['ms_per_step']))
PY
