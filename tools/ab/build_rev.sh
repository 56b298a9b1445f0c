#!/bin/bash
# builds build_tmp/lib<rev>.so: the product library from the sources of a git revision (bisecting a kernel regression against the same Python):
#   tools/ab/build_rev.sh <rev>;  then  SLIDE_HIP_LIB=$PWD/build_tmp/lib<rev>.so python ...
set -e
REV=$1
cd "$(dirname "$0")/../.."
O=build_tmp/rev_$REV; rm -rf $O; mkdir -p $O/slide_amd/csrc $O/include; C=$O/slide_amd/csrc
for f in $(git ls-tree --name-only $REV slide_amd/csrc/ | grep -E "\.(hip|h)$"); do git show $REV:$f > $C/$(basename $f); done
for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > $O/include/$(basename $f); done
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -I $C"
U="-mllvm -pragma-unroll-threshold=100000"
/opt/rocm/bin/hipcc $F -ffp-contract=off -c $C/point_ops.hip -o $O/point_ops.o 2>/dev/null &
for f in engine gemm_gx gemm_gxs point_chain; do /opt/rocm/bin/hipcc $F $U -c $C/$f.hip -o $O/$f.o 2>/dev/null & done
for f in rows_ops train_ops; do /opt/rocm/bin/hipcc $F -c $C/$f.hip -o $O/$f.o 2>/dev/null & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tmp/lib$REV.so $O/*.o
ls -la build_tmp/lib$REV.so
