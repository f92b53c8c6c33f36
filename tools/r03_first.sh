#!/bin/bash
# round-3 opening measurements on one GPU box: the any-order launch probe, the GPU tests, the baseline bench lines,
# re-measurement of two slab-size variants, the two-stream step with both graph replay modes -> gpurun_out/r03/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
timeout -k 10 120 ./scratch/anyorder_probe > $OUT/anyorder_probe.txt 2>&1; echo "probe exit=$?"; cat $OUT/anyorder_probe.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -15 $OUT/gpu_tests.log
cp gpurun_out/parity_errors.json $OUT/parity_errors_first.json 2>/dev/null
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines_first.jsonl; echo "$label done"; }
run "driver-style (20 steps, 5 warm-up)" timeout -k 10 300 python bench.py --steps 20 --warmup 5
run "c2 steady state (200 steps)" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run "two streams in one graph, direct launch" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams 2 --no-extras
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "two streams in one graph, packet capture" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams 2 --no-extras
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
bash tools/ab.sh ab_slabs.txt --steps 200 --warmup 50 -- main bwd3 fwd2
