"""slide_amd -- MI355X-native latent-DDPM sampling hot path of SLIDE (HIP kernels behind a C-ABI)."""
import os as _os

# Kernel arguments in device memory (the ROCm default): with HIP_FORCE_DEV_KERNARG=0 the launch-bound step plans of this
# package run 10 % slower (363 vs 406 shapes/s, bench.py on one MI355X).  Read by the HIP runtime when it initialises, so it
# only has an effect if nothing has touched the GPU yet; an explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
