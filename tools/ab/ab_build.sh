#!/bin/bash
# builds build_tmp/libA.so from the engine.hip of a git revision (default HEAD) for A/B timing against the working tree:
#   tools/ab/ab_build.sh [rev];  then on the GPU box:  SLIDE_HIP_LIB=$PWD/build_tmp/libA.so python tools/ab/time_chains.py
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
mkdir -p build_tmp/a/slide_amd/csrc build_tmp/a/include
git show $REV:slide_amd/csrc/engine.hip > build_tmp/a/slide_amd/csrc/engine.hip
git show $REV:slide_amd/csrc/point_ops.hip > build_tmp/a/slide_amd/csrc/point_ops.hip
for f in slide_hip.h slide_engine.h; do git show $REV:include/$f > build_tmp/a/include/$f; done
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Ibuild_tmp/a/include"
/opt/rocm/bin/hipcc $F -ffp-contract=off -c build_tmp/a/slide_amd/csrc/point_ops.hip -o build_tmp/a/point_ops.o 2>/dev/null
/opt/rocm/bin/hipcc $F -mllvm -pragma-unroll-threshold=100000 -c build_tmp/a/slide_amd/csrc/engine.hip -o build_tmp/a/engine.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/libA.so build_tmp/a/point_ops.o build_tmp/a/engine.o
ls -la build_tmp/libA.so
