#!/bin/bash
# usage: scripts/exp_build.sh <name> "<extra flags>" [source]  -> texture-gs_amd/libtexgs_<name>.so from an experimental render source
# (default build_exp/exp/render_exp.hip) + the product's other objects.  Experiments only; nothing here ships.
set -e
cd /root/repo
NAME=$1; FLAGS=$2; SRC=${3:-build_exp/exp/render_exp.hip}
mkdir -p build_exp/obj_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Itexture-gs_amd/csrc -Wall -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize $FLAGS -c $SRC -o build_exp/obj_$NAME/render.o
OBJS=$(ls texture-gs_amd/build/*.o | grep -v render.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o texture-gs_amd/libtexgs_$NAME.so build_exp/obj_$NAME/render.o $OBJS
echo built texture-gs_amd/libtexgs_$NAME.so
