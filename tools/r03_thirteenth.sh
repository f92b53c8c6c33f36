#!/bin/bash
# backward slab kernel: H-pass with a sliver of the halo plane per thread (committed) against the second item for 128 threads
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03e; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3; do
  for v in product bsold; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/gaps.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/gaps.txt
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v$rep -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/$v$rep.log 2>&1; echo "$v $rep exit=$?"
    python3 tools/trace_gaps.py $OUT/$v$rep | grep -v "^gap" >> $OUT/gaps.txt
  done
done
cat $OUT/gaps.txt
