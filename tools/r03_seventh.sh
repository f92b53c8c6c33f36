#!/bin/bash
# round 3: forward slab kernel variants (slabs per workgroup, planes per slab) on one box -> gpurun_out/r03b/ab2.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab2.txt
for rep in 1 2; do
  for v in "" one zs2r4 zs2r2; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab2.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v amdgpu.ids >> $OUT/ab2.txt || exit 1
  done
done
cat $OUT/ab2.txt
