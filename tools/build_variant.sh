#!/bin/bash
# Build a variant of libdpc_render.so into scratch/<name>/ with extra compiler flags (timing experiments, stamps):
#   tools/build_variant.sh <name> [extra hipcc flags...]        e.g.  tools/build_variant.sh abl -DDPC_ABLATE
# The variant is loaded by pointing dpc.render._native.LIB_PATH (or DPC_RENDER_LIB) at it; never shipped as the product.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/pytorch-unsup-pc_amd/csrc
out=$root/scratch/$name
mkdir -p $out
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=400000 $*"
pids=()
for f in dpc_slab_fwd dpc_slab_xl dpc_column dpc_slab_bwd dpc_entry dpc_stages dpc_nearest dpc_profile; do
  extra=""; [ $f = dpc_nearest ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $FLAGS $extra -c $src/$f.hip -o $out/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/libdpc_render.so
rm -f $out/*.o
echo "built $out/libdpc_render.so"
