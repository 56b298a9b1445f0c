"""gpurun_out/dechbm_<tag>/ (tools/decode_hbm.sh) -> profiles/<tag>_decode_hbm_pmc.md: per-kernel HBM-side bytes (FETCH x2 gfx950
correction + WRITE, KiB as rocprofv3 reports them) and the rate they move at, for the decode leg with fp16 operands."""
import collections, csv, glob, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join("gpurun_out", "dechbm_" + tag)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN12_GLOBAL__N_1"):
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
        n = (m.group(1) if m else n) + ("<f16>" if "DF16_" in n else "")
    return re.sub(r"\(.*", "", n)[:64]


def pmc(path, name):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


fa = pmc(glob.glob(os.path.join(src, "fetch", "**", "*counter_collection.csv"), recursive=True)[0], "FETCH_SIZE")
wa = pmc(glob.glob(os.path.join(src, "write", "**", "*counter_collection.csv"), recursive=True)[0], "WRITE_SIZE")
log = [l for l in open(os.path.join(src, "fetch.log")).read().splitlines() if "shapes/s" in l]
tot_b = tot_us = 0.0
rows = []
for k in set(fa) | set(wa):
    f, w = fa.get(k, [0, 0, 0]), wa.get(k, [0, 0, 0])
    kib = 2 * f[0] + w[0]
    us = f[2] if f[1] else w[2]
    rows.append((k, max(f[1], w[1]), kib * 1024 / 1e6, us / 1e3, (kib * 1024 / (us * 1e-6) / 1e9) if us else 0.0))
    tot_b += kib * 1024; tot_us += us
rows.sort(key=lambda r: -r[3])
with open(os.path.join("profiles", tag + "_decode_hbm_pmc.md"), "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `SLIDE_MODULE_PREC=fp16 python tools/time_decode.py` "
            "(the whole process: warm-up + timed decode passes of 256 latents)\n\n")
    o.write("under the profiler: `%s`\n\n" % (log[-1] if log else ""))
    o.write("HBM-side MB = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, "
            "MI355X_MICROARCH.md); GB/s = those bytes / the kernel's summed duration.  **All kernels: %.0f MB in %.1f ms of kernel time = "
            "%.0f GB/s = %.2f of the 8 TB/s peak.**\n\n" % (tot_b / 1e6, tot_us / 1e3, tot_b / (tot_us * 1e-6) / 1e9, tot_b / (tot_us * 1e-6) / 8e12))
    o.write("| kernel | dispatches | HBM-side MB | kernel ms | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|\n")
    for k, n, mb, ms, gbs in rows[:24]:
        o.write("| %s | %d | %.0f | %.2f | %.0f | %.2f |\n" % (k, n, mb, ms, gbs, gbs / 8000.0))
print("all kernels: %.0f MB, %.1f ms, %.0f GB/s" % (tot_b / 1e6, tot_us / 1e3, tot_b / (tot_us * 1e-6) / 1e9))
