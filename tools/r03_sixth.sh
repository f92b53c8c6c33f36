#!/bin/bash
# round 3: forward slab kernel that stays for several slabs, against the one-slab build on the same box -> gpurun_out/r03b/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab.txt $OUT/bench_lines.jsonl
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests exit=$?"; tail -3 $OUT/gpu_tests.log
for rep in 1 2; do
  for v in "" one; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    [ "$v" = abl ] && [ $rep = 2 ] && continue
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python tools/bench_step.py 400 >> $OUT/ab.txt 2>&1 || exit 1
  done
done
cat $OUT/ab.txt
run() { label=$1; shift; line=$("$@" 2>>$OUT/bench.err | tail -1); echo "{\"label\": \"$label\", \"line\": $line}" >> $OUT/bench_lines.jsonl; echo "$label done: $(echo $line | cut -c1-200)"; }
run "driver-style default" timeout -k 10 300 python bench.py --no-cpu-baseline
run "c2 steady" timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline
run "c5" timeout -k 10 300 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline
run "c4" timeout -k 10 300 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline
