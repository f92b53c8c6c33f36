#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03o; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product wscalar; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
unset DPC_RENDER_LIB
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "config4 or 128 or several_slabs or sigma or golden or fused" 2>&1 | tail -3
