#!/bin/bash
# Build a variant of libdpc_render.so into scratch/<name>/ with extra compiler flags (timing experiments, stamps):
#   tools/build_variant.sh <name> [extra hipcc flags...]        e.g.  tools/build_variant.sh abl -DDPC_ABLATE
#   SRC_REV=<git rev> tools/build_variant.sh <name> ...         builds the sources of that commit instead of the working tree
# The variant is loaded by pointing dpc.render._native.LIB_PATH (or DPC_RENDER_LIB) at it; never shipped as the product.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/pytorch-unsup-pc_amd/csrc
out=$root/scratch/$name
mkdir -p $out
if [ -n "$SRC_REV" ]; then   # the sources of a commit, laid out like the tree (csrc includes ../../include/dpc_render.h)
  tmp=$out/src; rm -rf $tmp; mkdir -p $tmp/pytorch-unsup-pc_amd/csrc $tmp/include
  for f in $(git -C $root ls-tree --name-only $SRC_REV pytorch-unsup-pc_amd/csrc/ include/); do git -C $root show $SRC_REV:$f > $tmp/$f; done
  src=$tmp/pytorch-unsup-pc_amd/csrc
fi
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=400000 $*"
pids=()
for f in dpc_slab_fwd dpc_slab_xl dpc_column dpc_slab_bwd dpc_entry dpc_stages dpc_nearest dpc_profile; do
  extra=""; [ $f = dpc_nearest ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $FLAGS $extra -c $src/$f.hip -o $out/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o $out/libdpc_render.so
rm -rf $out/*.o $out/src
echo "built $out/libdpc_render.so"
