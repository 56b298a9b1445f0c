#!/bin/bash
# builds build_tmp/lib<name>.so: the product library's sources with EXTRA compiler flags, for A/B timing against the in-tree build
# (SLIDE_HIP_LIB=build_tmp/lib<name>.so; tools/ab/ab_lib.sh).  build_tmp/ is git-ignored and travels with gpurun.
#   tools/ab/build_variant.sh noslp -fno-slp-vectorize
#   tools/ab/build_variant.sh maxilp -mllvm -amdgpu-sched-strategy=max-ilp
set -e
NAME=$1; shift
cd "$(dirname "$0")/../.."
O=build_tmp/$NAME; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -I slide_amd/csrc -Rpass-analysis=kernel-resource-usage $*"
U="-mllvm -pragma-unroll-threshold=100000"
/opt/rocm/bin/hipcc $F -ffp-contract=off -c slide_amd/csrc/point_ops.hip -o $O/point_ops.o 2> $O/point_ops.rem &
for f in engine gemm_gx gemm_gxs point_chain; do /opt/rocm/bin/hipcc $F $U -c slide_amd/csrc/$f.hip -o $O/$f.o 2> $O/$f.rem & done
for f in rows_ops train_ops; do /opt/rocm/bin/hipcc $F -c slide_amd/csrc/$f.hip -o $O/$f.o 2> $O/$f.rem & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/lib$NAME.so $O/*.o
ls -la build_tmp/lib$NAME.so
echo "spills / scratch:"; grep -h "Spill: [1-9]\|ScratchSize.*: [1-9]" $O/*.rem | sort | uniq -c | head
