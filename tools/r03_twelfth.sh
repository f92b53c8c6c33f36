#!/bin/bash
# rocprofv3 kernel trace of the c2 step with the committed library and with the one-slab build, same box, alternating
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03c; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product one; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v$rep -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/$v$rep.log 2>&1; echo "$v $rep exit=$?"
    echo "== $v rep $rep" >> $OUT/gaps.txt; python3 tools/trace_gaps.py $OUT/$v$rep >> $OUT/gaps.txt
  done
done
cat $OUT/gaps.txt
