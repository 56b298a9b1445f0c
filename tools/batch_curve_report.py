"""profiles/<tag>_batch_curve.md from gpurun_out/prof_<tag>_curve (tools/rocprof_run6.sh): the one-GPU batch curve of the default
arrangement -- shapes/s and time per 256 shapes against the per-GPU batch (bench lines outside the profiler), the dominant kernels under
rocprofv3 --kernel-trace --stats at each batch, and the MFMA counters of one feature step at the matching samples per launch."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06a"
src = os.path.join("gpurun_out", "prof_%s_curve" % tag)
out = os.path.join("profiles", "%s_batch_curve.md" % tag)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+)", n)
        n = (m.group(1) if m else n)
    return n.split("(")[0][:60]


L = ["# One-GPU batch curve of the default arrangement (position DDPM split + feature DDPM fp16, three feature sub-batches + the position chain)", "",
     "`python bench.py --gpus 1 --batch <B> --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs` (eager replay, "
     "outside the profiler); kernel shares from `rocprofv3 --kernel-trace --stats` of the same arrangement in graph replay (60 steps); MFMA counters from "
     "`rocprofv3 --pmc ... -- python tools/profile_ops.py --batch <samples per launch>` (one feature step, every launch alone on the GPU).", "",
     "## Throughput against the per-GPU batch", "",
     "| per-GPU batch | feature sub-batches | shapes/s | ms per joint step | ms per 256 shapes | vs batch 256 | host enqueue ms/step |", "|---|---|---|---|---|---|---|"]
base = None
for b in (256, 512, 1024, 2048):
    f = os.path.join(src, "bench_%d.json" % b)
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    per256 = d["ms_per_step"] * 256.0 / b
    base = base or per256
    L.append("| %d | %s | %.1f | %.4f | %.4f | %.3f x | %.3f |" % (b, "/".join(str(v) for v in d["config"]["sub_batches"]), d["value"], d["ms_per_step"], per256,
                                                              base / per256, d["config"]["host_enqueue_ms_per_step"]))
L += ["", "The batch-2048 row is what the kernels deliver when every launch holds several waves of workgroups and three chains overlap whatever is "
      "left: the resource-bound rate of the CURRENT kernels.  It bounds what any restructuring of the launch sequence at batch 256 (per-sample dataflow, "
      "persistent stages) could reach without making the kernels themselves faster.", ""]
L += ["## Kernel time shares at each batch (rocprofv3 --kernel-trace --stats, graph replay)", ""]
for b in (256, 512, 1024, 2048):
    fs = glob.glob(os.path.join(src, "trace_%d" % b, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    L += ["### per-GPU batch %d" % b, "", "| kernel | calls | avg us | % of kernel time |", "|---|---|---|---|"]
    for r in rows[:10]:
        L.append("| %s | %s | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot))
    L.append("")
L += ["## MFMA counters of one feature step against the samples per launch", "",
      "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", ""]
names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAIT_ANY"]
table = collections.OrderedDict()
sizes = []
for n in (88, 176, 344, 688):
    fs = glob.glob(os.path.join(src, "mfma_%d" % n, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    sizes.append(n)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k in acc:
        if not any(t in k for t in ("gemm", "attn", "sa_chain", "pair_first", "point_chain")) or cnt[k] == 0:
            continue
        a = acc[k]
        table.setdefault(k, {})[n] = (dur[k] / cnt[k], a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(dur[k] * 2400.0 * 1024.0, 1),
                                      a["SQ_INSTS_VALU"] / max(a["SQ_INSTS_MFMA"], 1), a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1))
L.append("| kernel | " + " | ".join("%d: avg us / MFMA util / VALU per MFMA / wait" % n for n in sizes) + " |")
L.append("|---|" + "---|" * len(sizes))
for k, v in sorted(table.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
    L.append("| %s | " % k + " | ".join(("%.1f / %.3f / %.1f / %.2f" % v[n]) if n in v else "-" for n in sizes) + " |")
L.append("")
for n in sizes:
    f = os.path.join(src, "ops_%d.txt" % n)
    if os.path.exists(f):
        t = [l for l in open(f) if l.startswith("total us")]
        if t:
            us = float(t[-1].split()[2])
            L.append("* one feature step alone at %d samples per launch (event-timed launches, tools/profile_ops.py): %.1f us = %.2f us per sample" % (n, us, us / n))
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
