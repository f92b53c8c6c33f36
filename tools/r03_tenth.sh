#!/bin/bash
# round 3: forward slab kernels with records fetched one slab ahead, against the previous commit -> gpurun_out/r03b/ab5.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab5.txt
for rep in 1 2; do
  for v in "" prev; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab5.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v amdgpu.ids >> $OUT/ab5.txt || exit 1
    for c in "--config c4" "--config c5" "--config c2 --sigma-rel 3.0"; do
      DPC_RENDER_LIB=$lib timeout -k 10 200 python bench.py $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab5.txt || echo "$c failed" >> $OUT/ab5.txt
    done
  done
done
cat $OUT/ab5.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests exit=$?"; tail -3 $OUT/gpu_tests.log
