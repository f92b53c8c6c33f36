#!/bin/bash
# GPU box: build and run tools/microbench/sc1_b128_store_probe.hip for tap radius 1 and 3 (three store shapes compared value by value)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/sc1_probe
for rb in 1 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRUN -DRB=$rb tools/microbench/sc1_b128_store_probe.hip -o gpurun_out/sc1_probe/probe_rb$rb 2>&1 | grep -i error
  timeout -k 5 60 gpurun_out/sc1_probe/probe_rb$rb | tee -a gpurun_out/sc1_probe/result.txt
  rm -f gpurun_out/sc1_probe/probe_rb$rb
done
