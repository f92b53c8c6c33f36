#!/bin/bash
# round 3: k_splat_hw staying for several slabs (c4, c3 at radius 10) against the one-slab build -> gpurun_out/r03b/ab4.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b; mkdir -p $OUT; rm -f $OUT/ab4.txt
for rep in 1 2; do
  for v in "" hw1; do
    lib=""; [ -n "$v" ] && lib=$PWD/scratch/$v/libdpc_render.so
    echo "== variant '${v:-product}' rep $rep" >> $OUT/ab4.txt
    DPC_RENDER_LIB=$lib timeout -k 10 200 python bench.py --config c4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value']), d['ms_per_step'], {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()} if 'all_kernels' in d.get('roofline',{}) else '')" >> $OUT/ab4.txt || exit 1
    DPC_RENDER_LIB=$lib timeout -k 10 200 python bench.py --config c2 --sigma-rel 3.0 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 sigma 3.0', round(d['value']), d['ms_per_step'])" >> $OUT/ab4.txt || echo "sigma run failed" >> $OUT/ab4.txt
  done
done
cat $OUT/ab4.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests exit=$?"; tail -3 $OUT/gpu_tests.log
