#!/bin/bash
# forward slab kernel with the next slab's records requested before this slab's (write-through) stores, against HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03q; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3; do
  for v in product prev; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/ab.txt
    timeout -k 10 200 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
DPC_RENDER_LIB=$PWD/scratch/abl/libdpc_render.so timeout -k 10 300 python tools/stamps.py c2 2>&1 | grep -v amdgpu.ids | head -12
unset DPC_RENDER_LIB
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
