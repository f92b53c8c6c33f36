#!/bin/bash
# write-through (sc1) stores of T and dT: dword stores of whole grid rows in the forward slab kernel, buffer stores with
# the sc1 bit in the column kernel -- against the committed library; then the whole GPU test run on the variant
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03j; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product ssc1d; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/ab.txt
    for c in "--config c5" "--config c4"; do
      timeout -k 10 200 python bench.py $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
export DPC_RENDER_LIB=$PWD/scratch/ssc1d/libdpc_render.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
