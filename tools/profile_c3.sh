#!/bin/bash
# rocprofv3 kernel statistics of the full training step (BASELINE configs[2]) as one HIP graph -> gpurun_out/prof_c3/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_c3; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config c3 --steps 30 --warmup 5 --captured "$@" > $OUT/trace.log 2>&1; echo "trace exit=$?"
python3 tools/profile_c3_summarise.py $OUT 30
tail -2 $OUT/trace.log | cut -c1-300
