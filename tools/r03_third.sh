#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $OUT/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -5 $OUT/gpu_tests.log
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines_third.jsonl; echo "$label done"; }
run "driver-style (20 steps, 5 warm-up), native step plan" timeout -k 10 300 python bench.py --steps 20 --warmup 5
run "driver-style again, no cpu" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run "c2 steady state (200 steps), plan" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline
run "c2 steady state (200 steps), graph" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --launch graph
run "c4 plan" timeout -k 10 300 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline
python - <<'PY'
import json
for l in open("gpurun_out/r03/bench_lines_third.jsonl"):
    d=json.loads(l); j=d['line']
    print(d['label'], round(j['value']), round(1e3*j['ms_per_step'],2), {k:round(v,2) for k,v in j['kernels_us'].items()}, j['roofline']['kernel'], round(j['roofline']['frac'],3), j['roofline_step'].get('own_traffic_frac'), j.get('hip_graph_replay',{}).get('median_us'), j.get('two_batches_in_flight',{}).get('point_clouds_per_sec'))
PY
