#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
timeout -k 10 120 ./scratch/xstream_probe > $OUT/xstream_probe.txt 2>&1; echo "probe exit=$?"; cat $OUT/xstream_probe.txt
DPC_RENDER_LIB=$PWD/scratch/abl/libdpc_render.so timeout -k 10 300 python tools/stamps.py c2 > $OUT/stamps_c2.txt 2>&1; echo "stamps exit=$?"; cat $OUT/stamps_c2.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -25 $OUT/gpu_tests.log
cp gpurun_out/parity_errors.json $OUT/parity_errors_second.json 2>/dev/null
