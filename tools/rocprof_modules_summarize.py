"""gpurun_out/profm_<tag>/ (tools/rocprof_modules.sh) -> profiles/<tag>_<run>_kernel_stats.md"""
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join("gpurun_out", "profm_" + tag)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:80]


for run in ("decode", "decode_fp16", "encode", "encode_fp16", "train32", "train256"):
    f = glob.glob(os.path.join(src, run, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    log = [l for l in open(os.path.join(src, run + ".log")).read().splitlines() if "shapes/s" in l or "clouds/s" in l or "samples/s" in l]
    with open(os.path.join("profiles", "%s_%s_kernel_stats.md" % (tag, run)), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats: module-level path, %s (%s)\n\n" % (run, "tools/time_train.py" if run.startswith("train") else "tools/time_%s.py" % run.split("_")[0]))
        if run.startswith("train"):
            o.write("(`tools/time_train.py %s --eager-only`: the eager training step of both denoisers; the graphed step replays the same kernels)\n\n" % run[5:])
        o.write("under the profiler: `%s`\n\n" % (" | ".join(log[-2:]) if run.startswith("train") else (log[-1] if log else "")))
        o.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:(30 if run.startswith("train") else 22)]:
            o.write("| %s | %s | %.2f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print(run, log[-1] if log else "")
