#!/bin/bash
# round 3: backward slab kernel variants (H-pass tail, record-ahead) on one box -> gpurun_out/r03b/ab3.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab3.txt
for rep in 1 2; do
  for v in "" g0 g1; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab3.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v amdgpu.ids >> $OUT/ab3.txt || exit 1
  done
done
DPC_RENDER_LIB=$PWD/scratch/abl/libdpc_render.so timeout -k 10 300 python tools/stamps.py c2 2>&1 | grep -v amdgpu.ids >> $OUT/ab3.txt
cat $OUT/ab3.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests exit=$?"; tail -3 $OUT/gpu_tests.log
