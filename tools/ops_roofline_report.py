"""Builds profiles/<tag>_ops_roofline.md from rocprofv3 runs of tools/ops_roofline.py:
   <dir>/trace (kernel trace), <dir>/fetch (--pmc FETCH_SIZE), <dir>/write (--pmc WRITE_SIZE), <dir>/manifest.json.
Dispatches are attributed to cases through the marker kernel (arange) between cases; per case the kernel time is the MEDIAN
duration of this repo's kernel(s) of one call.  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md section HBM:
it reports 1/2 of a wide coalesced stream); both counters are in KB."""
import csv
import re, glob, json, os, sys
import numpy as np

d, out = sys.argv[1], sys.argv[2]
man = json.load(open(os.path.join(d, "manifest.json")))["cases"]
OURS = ("group_points", "gather_points", "gather_rows", "three_interpolate", "three_nn", "ball_query", "knn", "fps_kernel", "sample_farthest", "furthest")


def split_by_marker(rows, key):
    groups, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "arange" in name:
            if cur is not None:
                groups.append(cur)
            cur = []
        elif cur is not None and any(k in name for k in OURS):
            cur.append(r)
    return groups


def load(sub, pattern):
    f = glob.glob(os.path.join(d, sub, "**", pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


tr = load("trace", "*kernel_trace.csv")
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
gt = split_by_marker(tr, None)
pm = {}
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = [r for r in load(sub, "*counter_collection.csv")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # one row per (dispatch, counter): keep the wanted counter, but markers carry it too
    rows = [r for r in rows if r["Counter_Name"] == cname]
    pm[cname] = split_by_marker(rows, None)
assert len(gt) >= len(man), (len(gt), len(man))
lines = ["| op | shape | kernel | us / call | algorithmic MB (r + w) | GB/s | frac of 8 TB/s | FETCHx2 + WRITE MB | moved GB/s (frac) | note |", "|---|---|---|---|---|---|---|---|---|---|"]
js = []
for i, c in enumerate(man):
    g_ = gt[i]
    per_call = max(1, len(g_) // c["reps"])
    dur = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g_], np.float64).reshape(c["reps"], per_call).sum(1) / 1e3
    us = float(np.median(dur))
    m_ = re.search(r"(\w+)(?:<[^(]*>)?\(", g_[0]["Kernel_Name"].replace("(anonymous namespace)::", "")) if g_ else None
    name = m_.group(1) if m_ else next((k for k in ("three_interpolate_rows_kernel",) if g_ and k in g_[0]["Kernel_Name"]), "?")
    alg = c["read_B"] + c["write_B"]
    traffic = None
    if pm["FETCH_SIZE"] and pm["WRITE_SIZE"] and i < len(pm["FETCH_SIZE"]) and i < len(pm["WRITE_SIZE"]):
        f_ = np.median(np.array([float(r["Counter_Value"]) for r in pm["FETCH_SIZE"][i]]).reshape(c["reps"], -1).sum(1))
        w_ = np.median(np.array([float(r["Counter_Value"]) for r in pm["WRITE_SIZE"][i]]).reshape(c["reps"], -1).sum(1))
        traffic = (2 * f_ + w_) * 1024 / 1e6
    note = ""
    if "dist_evals" in c:
        note = "%.2f T dist/s" % (c["dist_evals"] / us / 1e6)
    if "selections" in c:
        note = "%.2f us / selection" % (us / c["selections"])
    if "physical_read_B" in c:
        # (B, C, N) layout: a touched element costs its whole 64-byte sector, and with m >= n / 4 picked columns nearly every sector of a
        # channel row holds one -- the kernel cannot move fewer bytes than the rows it touches + what it writes (the LAYOUT FLOOR);
        # `frac` on algorithmic bytes is bounded by algorithmic / floor whatever the kernel does
        floor = c["physical_read_B"] + c["write_B"]
        note = "layout floor %.0f MB (rows touched %.0f + written): %.2f of 8 TB/s on it; algorithmic frac <= %.2f" % (
            floor / 1e6, c["physical_read_B"] / 1e6, floor / us / 1e3 / 8000.0, alg / floor)
    gbs = alg / us / 1e3
    moved = "-" if traffic is None else "%.0f (%.2f)" % (traffic / us * 1e3, traffic / us / 8.0)
    lines.append("| %s | %s | `%s` | %.1f | %.1f | %.0f | %.3f | %s | %s | %s |" % (c["op"], c["shape"], name, us, alg / 1e6, gbs, gbs / 8000.0,
                                                                               "%.1f" % traffic if traffic is not None else "-", moved, note))
    js.append(dict(op=c["op"], shape=c["shape"], us=round(us, 2), algorithmic_MB=round(alg / 1e6, 2), GBps=round(gbs, 1),
                   frac=round(gbs / 8000.0, 4), traffic_MB=None if traffic is None else round(traffic, 1), note=note))
open(out, "w").write("\n".join(lines) + "\n")
json.dump(js, open(out.replace(".md", ".json"), "w"), indent=1)
print("\n".join(lines))
