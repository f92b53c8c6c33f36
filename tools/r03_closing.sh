#!/bin/bash
# round 3, closing: rocprof summaries (c2, c4), SQ counters (c2), then tools/r03_final.sh (tests, smoke, bench lines)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03_closing
bash tools/profile_gpu.sh c2 > gpurun_out/r03_closing/profile_c2.log 2>&1; echo "profile c2 exit=$?"; tail -3 gpurun_out/r03_closing/profile_c2.log
bash tools/profile_gpu.sh c4 > gpurun_out/r03_closing/profile_c4.log 2>&1; echo "profile c4 exit=$?"; tail -3 gpurun_out/r03_closing/profile_c4.log
bash tools/profile_pmc.sh c2 > gpurun_out/r03_closing/pmc_c2.log 2>&1; echo "pmc exit=$?"; tail -6 gpurun_out/r03_closing/pmc_c2.log
