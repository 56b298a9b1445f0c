#!/bin/bash
# A/B of two builds inside one gpurun call: build_tmp/libA.so (the previous build, copied before rebuilding) vs the tree's
for i in 1 2 3; do
  for L in "" "$PWD/build_tmp/libA.so"; do
    echo -n "${L:-tree}: "; SLIDE_HIP_LIB=$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | cut -c70-100
  done
done
