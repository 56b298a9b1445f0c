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
           ("gemm_chain.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]),
           ("resident.hip", ["-mllvm", "-pragma-unroll-threshold=100000"]), ("rows_ops.hip", [])]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function"]


# Kernels that are allowed to spill: nothing a DEFAULT path dispatches.  Every entry is an opt-in /
# fallback instantiation (the knob that selects it in brackets); any other kernel with a spilled VGPR or a private segment
# fails the build.  Patterns are regular expressions over the demangled kernel name.
SPILL_OPT_IN = [
    r"gemm_glds_kernel<\d, 4, ",            # 128-channel ring tiles [SLIDE_CBW4_TILES]
    r"gemm_glds_kernel<\d, \d, 3, 64, ",    # 64-deep chunks [SLIDE_GLDS_WIDE=1]
    r"gemm_glds_kernel<7, 2, 3, 32, true",  # round-2 FP key GEMM [SLIDE_GX=0]
    r"gemm_glds_occ3_kernel<7, ",           # round-2 FP blocks [SLIDE_GX=0]
    r"gemm_glds_kernel<4, 2, ",             # 256-row ring tiles over 16-row samples [SlideOp.i[9] == 3, > 8192 small tiles]
    r"gemm_kernel<1, ",                     # register-staged fp16 GEMM [SLIDE_GLDS=0]
    r"gemm_xs_kernel<",                     # X-stationary GEMM [SLIDE_XS]
    r"resident_kernel",                     # LDS-resident position denoiser [ResidentPositionSampler]
    r"pair_norm2_kernel",                   # one-workgroup-per-sample table pass [SLIDE_PAIR_NORM_V2=1]
    r"block_body_kernel<8, ",               # SA0 block body [SLIDE_BODY=2]
]


def parse_resource_remarks(stderr):
    """hipcc -Rpass-analysis=kernel-resource-usage -> [(mangled name, vgprs, agprs, spills, scratch, lds, waves/SIMD)] (device pass)"""
    import re
    rows, cur = [], None
    for line in stderr.split("\n"):
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|"
                      r"Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k not in cur:
            cur[k] = int(v)
    return [(r["name"], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("VGPRs Spill", 0), r.get("ScratchSize [bytes/lane]", 0),
             r.get("LDS Size [bytes/block]", 0), r.get("Occupancy [waves/SIMD]", 0)) for r in rows]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for o in out[:len(names)]]


def check_spills(src, stderr):
    """fails the build when a kernel outside SPILL_OPT_IN spills; returns the resource rows of the source"""
    import re
    rows = parse_resource_remarks(stderr)
    names = demangle([r[0] for r in rows])
    rows = [(n,) + r[1:] for n, r in zip(names, rows)]
    bad = [r for r in rows if (r[3] or r[4]) and not any(re.search(p, r[0]) for p in SPILL_OPT_IN)]
    if bad:
        raise RuntimeError("%s: kernels on a default path spill registers:\n%s" % (
            src, "\n".join("  %s: %d VGPRs spilled, %d B scratch" % (r[0], r[3], r[4]) for r in bad)))
    return rows


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
            cmd = [HIPCC] + COMMON + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            try:
                rows = check_spills(src, r.stderr)
            except RuntimeError:
                os.remove(o)  # (the next build re-checks instead of linking the stale object)
                raise
            with open(o + ".resources", "w") as f:  # tools/kernel_resources.py collects these into profiles/
                for row in rows:
                    f.write("\t".join(str(v) for v in row) + "\n")
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
