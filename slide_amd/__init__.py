"""slide_amd -- MI355X-native latent-DDPM sampling hot path of SLIDE (HIP kernels behind a C-ABI)."""
import os as _os

# Kernel arguments in device memory (the ROCm default): with HIP_FORCE_DEV_KERNARG=0 the launch-bound step plans of this
# package run 10 % slower (363 vs 406 shapes/s, bench.py on one MI355X).  Read by the HIP runtime when it initialises, so it
# only has an effect if nothing has touched the GPU yet; an explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

# Hardware queues per process (the ROCm default is 4): the sampling arrangement replays FOUR chains (the position chain + three feature
# sub-batches) on four streams, and two chains that land on one hardware queue serialise -- measured with bench.py (round 5,
# tools/ab/r05_queues.sh): GPU_MAX_HW_QUEUES = 4 / 5 / 12 / 16 / 24 -> 385 shapes/s, 3 / 7 / 8 -> 285, 6 -> 320, 1 -> 195.  Pinned to the
# default unless the environment says otherwise (same init-time caveat as above).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
