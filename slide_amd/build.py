"""Builds the in-tree native libraries.

  slide_amd/libslide_hip.so   hand-written gfx950 HIP kernels + the C-ABI (include/slide_hip.h, slide_engine.h)
hipcc cross-compiles for gfx950 without a GPU, so this runs in the authoring container; the built .so
travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libslide_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (source, extra flags).  point_ops needs contraction OFF (bit-exact index parity with the oracle).
# engine: the GEMM epilogue's per-channel-block loop must unroll fully (the accumulators are indexed by it; left rolled
# they are demoted to scratch), which needs more than LLVM's default 16k-instruction cap for `#pragma unroll`.
SOURCES = [("point_ops.hip", ["-ffp-contract=off"]), ("engine.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]),
           ("gemm_xs.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]),
           ("gemm_gx.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]),
           ("block_body.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]),
           ("resident.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]), ("rows_ops.hip", [])]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    objs = []
    deps = [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    relink = force
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [HIPCC] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
        objs.append(o)
    if relink or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
