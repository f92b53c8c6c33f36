#!/bin/bash
# write-through stores are the product now (T, dT); knobs: k_splat_hw / k_zcol_bwd back to ordinary stores, k_locate's
# records and k_gather_hw's point gradients written through as well
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03k; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
  for v in product hwback zbback locthru dpcthru; do
    if [ $v = product ]; then unset DPC_RENDER_LIB; else export DPC_RENDER_LIB=$PWD/scratch/$v/libdpc_render.so; fi
    echo "== $v rep $rep" >> $OUT/ab.txt
    timeout -k 10 200 python tools/bench_step.py 400 2>&1 | grep -v "amdgpu.ids\|status word" >> $OUT/ab.txt
    for c in "--config c5" "--config c4" "--config c2 --api plain --launch graph"; do
      timeout -k 10 200 python bench.py $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v['avg_launch_us'],2) for k,v in d['roofline']['all_kernels'].items()})" >> $OUT/ab.txt
    done
  done
done
cat $OUT/ab.txt
unset DPC_RENDER_LIB
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
