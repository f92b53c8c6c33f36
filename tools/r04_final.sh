#!/bin/bash
# Round-4 closing session on one GPU box: the full GPU suite, smoke, and every bench line DESIGN.md quotes -> gpurun_out/r04_final/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/gpu.sh r04_final tests smoke \
  "bench:driver-style default (20 steps, 5 warm-up), with cpu_baseline:--steps 20 --warmup 5" \
  "bench:driver-style again:--steps 20 --warmup 5 --no-cpu-baseline" \
  "bench:c2 steady state (200 steps):--steps 200 --warmup 50 --no-cpu-baseline" \
  "bench:c2 as a replayed HIP graph of the autograd path:--steps 200 --warmup 50 --no-cpu-baseline --launch graph" \
  "bench:c2 through the reference's plain call sequence, replayed:--steps 100 --warmup 10 --no-cpu-baseline --api plain --launch graph" \
  "bench:c2 shapes at sigma_rel 3.0:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 3.0" \
  "bench:c2 shapes at sigma_rel 2.0:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 2.0" \
  "bench:c2 shapes at sigma_rel 1.5:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 1.5" \
  "bench:c2 shapes at sigma_rel 1.2:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 1.2" \
  "bench:c2 shapes at sigma_rel 1.05:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 1.05" \
  "bench:c2 shapes at sigma_rel 0.9:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 0.9" \
  "bench:c2 shapes at sigma_rel 0.75:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 0.75" \
  "bench:c2 shapes at sigma_rel 0.5:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 0.5" \
  "bench:c2 shapes at sigma_rel 0.35:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 0.35" \
  "bench:c2 shapes at sigma_rel 0.2:--steps 200 --warmup 50 --no-cpu-baseline --no-extras --sigma-rel 0.2" \
  "bench:c4 shard:--config c4 --steps 100 --warmup 10 --no-cpu-baseline" \
  "bench:c5:--config c5 --steps 100 --warmup 10 --no-cpu-baseline" \
  "bench:c2 two ranks over gloo on one GPU (functional rehearsal of --gpus N):--gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu --no-cpu-baseline" \
  "bench:c3 full training step, all points, eager:--config c3 --steps 30 --warmup 5" \
  "bench:c3 full training step as one HIP graph, all points:--config c3 --steps 30 --warmup 5 --captured" \
  "bench:c3 full training step as one HIP graph, keep 0.07, schedules followed:--config c3 --steps 30 --warmup 5 --captured --keep 0.07" \
  "bench:c4 full training step (BASELINE configs[3] as worded), one rank, eager:--config c4 --full-step --steps 20 --warmup 3" \
  "bench:c4 full training step, one rank, one HIP graph:--config c4 --full-step --steps 20 --warmup 3 --captured"
