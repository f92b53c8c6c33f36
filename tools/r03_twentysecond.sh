#!/bin/bash
# cache policy of the two big LOAD streams: T in the column kernel (nt, sc1), dT in the backward slab kernel (nt)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03t; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product tlnt tlsc1 dtlnt; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
