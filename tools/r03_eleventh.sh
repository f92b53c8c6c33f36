#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab6.txt
for rep in 1 2 3; do
  for v in "" zearly; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab6.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v amdgpu.ids >> $OUT/ab6.txt || exit 1
    DPC_RENDER_LIB=$lib timeout -k 10 200 python bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab6.txt
  done
done
cat $OUT/ab6.txt
