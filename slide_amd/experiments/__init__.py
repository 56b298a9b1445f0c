"""Opt-in experiments: code paths that were built, parity-tested and A/B-timed but lost to the default plan.
They need the EXPERIMENTS build of the HIP library (`python slide_amd/build.py --experiments` -> libslide_hip_exp.so, loaded
with SLIDE_EXPERIMENTS=1 or inside `slide_amd._lib.experiments()`); the product library does not carry their kernels."""
