#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench command, then HBM traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes), plus a calibration run.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CONFIG=${1:-c2}
OUT=gpurun_out/prof_$CONFIG; rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --config $CONFIG --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace exit=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH --no-graph > $OUT/fetch.log 2>&1; echo "fetch exit=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --no-graph > $OUT/write.log 2>&1; echo "write exit=$?"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/hbm_calib.hip -o $OUT/hbm_calib 2>&1 | grep -i error
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $OUT/hbm_calib > $OUT/calib_fetch.log 2>&1; echo "calib fetch exit=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $OUT/hbm_calib > $OUT/calib_write.log 2>&1; echo "calib write exit=$?"
rm -f $OUT/hbm_calib
python3 tools/profile_summarise.py $OUT $CONFIG
