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
# EXPERIMENTS build (-DSLIDE_EXPERIMENTS): the product sources + every opt-in variant that lost its A/B (X-stationary GEMM,
# per-point layer chains, LDS-resident position denoiser, head + update launch, wide / eight-wave attention tails, 128- / 32-
# channel and 64-deep ring tiles, the round-2 plan's kernels, the register-staged fp16 GEMM, the SA0 block body, the two-launch
# pair-table pass).  Loaded instead of the product library with SLIDE_EXPERIMENTS=1 or `slide_amd._lib.experiments()`;
# the plan-variant tests use it.  Its kernels may spill; the PRODUCT library may not.
LIB_EXP = os.path.join(HERE, "libslide_hip_exp.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (source, extra flags).  point_ops needs contraction OFF (bit-exact index parity with the oracle).
# engine: the GEMM epilogue's per-channel-block loop must unroll fully (the accumulators are indexed by it; left rolled
# they are demoted to scratch), which needs more than LLVM's default 16k-instruction cap for `#pragma unroll`.
UNROLL = ["-mllvm", "-pragma-unroll-threshold=100000"]
SOURCES = [("point_ops.hip", ["-ffp-contract=off"]), ("engine.hip", UNROLL), ("gemm_gx.hip", UNROLL), ("gemm_gxs.hip", UNROLL),
           ("point_chain.hip", UNROLL), ("rows_ops.hip", []), ("train_ops.hip", [])]
SOURCES_EXP = SOURCES + [("experiments/block_body.hip", UNROLL), ("experiments/gemm_xs.hip", UNROLL), ("experiments/gemm_chain.hip", UNROLL),
                         ("experiments/resident.hip", UNROLL)]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function", "-I", CSRC]

# PRODUCT build: NO kernel may spill a register or use scratch (the build fails otherwise; SLIDE_ALLOW_SPILLS=1 downgrades that
# to a warning for other ROCm / LLVM versions).  The experiments build is not linted.
# The ONE exception: gemm_gx_dual_kernel<7> runs at three workgroups per CU (168 registers) with 15 values parked in scratch OUTSIDE
# its MFMA steps (gemm_gx.hip; measured +1.1 % on the headline against the spill-free two-per-CU form).
SPILL_OPT_IN = [r"gemm_gx_dual_kernel<7>"]


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
    """c++filt, else ROCm's llvm-cxxfilt; with neither the MANGLED names are returned (SPILL_OPT_IN patterns then match on
    the kernel's base name only: template arguments are encoded, so argument-specific patterns do not apply)"""
    import shutil
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "llvm-cxxfilt"):
        exe = shutil.which(tool) or (tool if os.path.isabs(tool) and os.path.exists(tool) else None)
        if exe is None:
            continue
        r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.split("\n")
        if r.returncode == 0 and len(out) >= len(names):
            return [o.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for o in out[:len(names)]]
    return list(names)


def check_spills(src, stderr):
    """fails the build when a kernel outside SPILL_OPT_IN spills; returns the resource rows of the source"""
    import re
    rows = parse_resource_remarks(stderr)
    names = demangle([r[0] for r in rows])
    rows = [(n,) + r[1:] for n, r in zip(names, rows)]
    bad = [r for r in rows if (r[3] or r[4]) and not any(re.search(p, r[0]) for p in SPILL_OPT_IN)]
    if bad:
        msg = "%s: kernels on a default path spill registers:\n%s" % (
            src, "\n".join("  %s: %d VGPRs spilled, %d B scratch" % (r[0], r[3], r[4]) for r in bad))
        if os.environ.get("SLIDE_ALLOW_SPILLS", "0") != "0":  # escape hatch for other ROCm / LLVM versions
            sys.stderr.write("WARNING (SLIDE_ALLOW_SPILLS=1): " + msg + "\n")
        else:
            raise RuntimeError(msg + "\n(set SLIDE_ALLOW_SPILLS=1 to build anyway)")
    return rows


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src, extra, obj, defines, lint, verbose):
    cmd = [HIPCC] + COMMON + extra + defines + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    # compiler diagnostics that are not resource remarks (-Wall warnings) are passed on, also on success
    diag = [ln for ln in r.stderr.split("\n") if ln.strip() and "remark:" not in ln and "[-Rpass-analysis" not in ln]
    if diag:
        sys.stderr.write("\n".join(diag) + "\n")
    rows = parse_resource_remarks(r.stderr)
    names = demangle([q[0] for q in rows])
    rows = [(n,) + q[1:] for n, q in zip(names, rows)]
    if lint:
        try:
            check_spills(os.path.basename(src), r.stderr)
        except RuntimeError:
            os.remove(obj)  # (the next build re-checks instead of linking the stale object)
            raise
    with open(obj + ".resources", "w") as f:  # tools/kernel_resources.py collects these into profiles/
        for row in rows:
            f.write("\t".join(str(v) for v in row) + "\n")


def _build_one(lib, sources, objdir, defines, lint, force, verbose):
    from concurrent.futures import ThreadPoolExecutor
    deps = [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    deps.append(os.path.abspath(__file__))
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src, extra in sources:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            jobs.append((s, extra, o))
        objs.append(o)
    if jobs:  # the translation units compile concurrently (hipcc is single-threaded per file; engine.hip is the long pole)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for f in [ex.submit(_compile, s, extra, o, defines, lint, verbose) for s, extra, o in jobs]:
                f.result()
    if jobs or not os.path.exists(lib):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False, experiments=False):
    """the product library; experiments=True: also (only with experiments="only") the experiments library"""
    if experiments != "only":
        _build_one(LIB, SOURCES, CSRC, [], True, force, verbose)
    if experiments:
        _build_one(LIB_EXP, SOURCES_EXP, os.path.join(CSRC, "exp"), ["-DSLIDE_EXPERIMENTS"], False, force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                experiments=("only" if "--experiments-only" in sys.argv else "--experiments" in sys.argv)))
