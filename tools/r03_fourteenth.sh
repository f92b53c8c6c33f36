#!/bin/bash
# column kernel with two rays per lane (packed FMAs, one wave per SIMD) against the committed one-ray form
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03f; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product rpl2w1 rpl2; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
export DPC_RENDER_LIB=$PWD/scratch/rpl2w1/libdpc_render.so
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fused or golden or step_plan or config2" 2>&1 | tail -3
