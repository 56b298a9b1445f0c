"""Turn gpurun_out/prof_<tag>/ (tools/rocprof_run.sh) into small tracked summaries under profiles/."""
import collections
import csv
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 88  # samples per launch of the PMC passes (tools/rocprof_run3.sh <tag> <batch>)
# optional third argument: the profiled command of the PMC passes when it is not the default feature step (round 5: the position
# DDPM's split plan, `--which pos --prec split`); hbm_pmc_latest.json (bench.py's `roofline.traffic`) is then left alone
pmc_cmd = sys.argv[3] if len(sys.argv) > 3 else None
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_]+)", n)
        n = (m.group(1) if m else n) + ("<f16>" if "DF16_" in n else "")
    return n.split("(")[0][:70]


rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats (%s): `python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --no-decode --replay graph` (graph replay: not host-bound under the profiler)\n\n" % tag)
    bl = os.path.join(src, "bench.log")
    if os.path.exists(bl):
        for line in open(bl):
            if line.startswith('{"metric"'):
                f.write("bench line under the profiler:\n\n```\n" + line.strip() + "\n```\n\n")
    f.write("| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:25]:
        f.write("| %s | %s | %.2f | %.1f | %s | %.1f | %.1f |\n" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"],
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))


def pmc(path, name):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


fp = os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv")
wp = os.path.join(src, "pmc_write", "write_counter_collection.csv")
if os.path.exists(fp) and os.path.exists(wp):
    fa, wa = pmc(fp, "FETCH_SIZE"), pmc(wp, "WRITE_SIZE")
    with open(os.path.join(dst, tag + "_hbm_pmc.md"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python tools/profile_ops.py --reps 2 --batch %d%s`\n\n" % (batch, " " + pmc_cmd if pmc_cmd else "") +
                "Units: KiB per dispatch as reported by rocprofv3.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x\n"
                "(MI355X_MICROARCH.md, HBM section): the `fetch x2` column applies that correction.\n\n"
                "| kernel | dispatches | FETCH KiB/disp | fetch x2 KiB | WRITE KiB/disp | avg us |\n|---|---|---|---|---|---|\n")
        keys = [k for k in fa if any(t in k for t in ("gemm", "assemble", "attn", "copy", "prep", "update", "finalize", "sa_chain", "block_body", "pair_norm", "pair_first", "head_update", "point_chain"))]
        for k in sorted(keys, key=lambda k: -fa[k][0]):
            w = wa.get(k, [0, 1, 0])
            f.write("| %s | %d | %.0f | %.0f | %.0f | %.1f |\n" % (k, fa[k][1], fa[k][0] / fa[k][1], 2 * fa[k][0] / fa[k][1],
                                                               w[0] / max(w[1], 1), fa[k][2] / fa[k][1]))
    import json
    kern = {}
    for k in fa:
        if any(t in k for t in ("gemm", "attn_tail", "sa_chain", "block_body", "pair_first", "pair_norm", "head_update", "point_chain")):
            w = wa.get(k, [0, 1, 0])
            kern[k] = {"hbm_bytes_per_launch": int(1024 * (2 * fa[k][0] / fa[k][1] + w[0] / max(w[1], 1))),
                       "fetch_kib_x2": 2 * fa[k][0] / fa[k][1], "write_kib": w[0] / max(w[1], 1), "dispatches": fa[k][1]}
    latest = os.path.join(dst, "hbm_pmc_latest.json")
    if pmc_cmd is not None and os.path.exists(latest):
        # (round 6) the position plan's kernels join the file under their own key, with their own launch size
        d_ = json.load(open(latest))
        d_["position_plan"] = {"source": "profiles/%s_hbm_pmc.md (`tools/profile_ops.py %s`; same passes and correction)" % (tag, pmc_cmd),
                               "samples_per_launch": batch, "kernels": kern}
        json.dump(d_, open(latest, "w"), indent=1)
    if pmc_cmd is None:
      json.dump({"source": "profiles/%s_hbm_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 gfx950 "
                         "correction; %d samples per launch)" % (tag, batch), "samples_per_launch": batch, "kernels": kern}, open(os.path.join(dst, "hbm_pmc_latest.json"), "w"), indent=1)
mp = os.path.join(src, "pmc_mfma", "mfma_counter_collection.csv")
if os.path.exists(mp):
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAIT_ANY"]
    acc = {n: pmc(mp, n) for n in names}
    with open(os.path.join(dst, tag + "_mfma_pmc.md"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY over "
                "`python tools/profile_ops.py --reps 2 --batch %d%s` (one %s step, every launch alone on the GPU)\n\n" % (batch, " " + pmc_cmd if pmc_cmd else "", "denoiser" if pmc_cmd else "feature-denoiser") +
                "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs): the fraction of the chip's "
                "SIMD-cycles with the matrix pipe busy while the kernel runs, priced at the MAXIMUM clock (a lower bound under DVFS; "
                "MI355X_MICROARCH.md: the counter advances 32 per 32x32x16 MFMA, summed over all SIMDs).  "
                "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of wave-cycles parked in s_waitcnt / barriers.\n\n"
                "| kernel | dispatches | avg us | MFMA utilisation | VALU per MFMA | wait share |\n|---|---|---|---|---|---|\n")
        ks = [k for k in acc["SQ_BUSY_CYCLES"] if any(t in k for t in ("gemm", "attn", "sa_chain", "block_body", "pair_norm", "pair_first", "point_chain"))]
        for k in sorted(ks, key=lambda k: -acc["SQ_BUSY_CYCLES"][k][2]):
            g_ = lambda n: acc[n].get(k, [0.0, 1, 0.0])[0]
            d = acc["SQ_BUSY_CYCLES"][k]
            f.write("| %s | %d | %.1f | %.3f | %.1f | %.2f |\n" % (k, d[1], d[2] / d[1], g_("SQ_VALU_MFMA_BUSY_CYCLES") / max(d[2] * 2400.0 * 1024.0, 1),
                                                                g_("SQ_INSTS_VALU") / max(g_("SQ_INSTS_MFMA"), 1), g_("SQ_WAIT_ANY") / max(g_("SQ_WAVE_CYCLES"), 1)))
print("written", sorted(os.listdir(dst)))
