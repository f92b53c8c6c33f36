#!/bin/bash
# round-2 baseline on a fresh box: tests, driver-style bench, phase stamps, SQ counters
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
python -m pytest tests -x -q -m gpu > gpurun_out/r02/gpu_tests.log 2>&1; echo "tests exit=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_drv1.json 2> gpurun_out/r02/bench_drv1.err; echo "bench1 exit=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_drv2.json 2>> gpurun_out/r02/bench_drv1.err; echo "bench2 exit=$?"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r02/bench_200.json 2>> gpurun_out/r02/bench_drv1.err; echo "bench3 exit=$?"
DPC_RENDER_LIB=$PWD/scratch/abl/libdpc_render.so python tools/stamps.py c2 > gpurun_out/r02/stamps_c2.log 2>&1; echo "stamps exit=$?"
bash tools/profile_pmc.sh > gpurun_out/r02/pmc.log 2>&1; echo "pmc exit=$?"
cp gpurun_out/pmc/summary.txt gpurun_out/r02/sq_counters.txt
