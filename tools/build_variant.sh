#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG=VALUE ...]   -> path-tracer_amd/lib/variants/libklrender_<name>.so
set -e
name=$1; shift
mkdir -p build/var_$name path-tracer_amd/lib/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -Iinclude -Ipath-tracer_amd/csrc $@"
for f in kl_kernels.hip kl_lbvh.hip kl_group.hip kl_api.cpp kl_bvh.cpp kl_scene.cpp kl_mesh_io.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c path-tracer_amd/csrc/$f -o build/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o path-tracer_amd/lib/variants/libklrender_$name.so build/var_$name/*.o -lpthread -ldl
echo built $name
