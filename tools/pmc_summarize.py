"""average rocprofv3 --pmc counter values per kernel: python tools/pmc_summarize.py gpurun_out/pmc_<tag> [filter]"""
import collections, csv, glob, sys
filt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if filt not in r["Kernel_Name"]:
            continue
        k = (r["Dispatch_Id"], r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:48], r["Grid_Size"])
        d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        d[k]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg = collections.OrderedDict()
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # dispatches per benchmark case (0 = group by kernel only)
    for n, (k, v) in enumerate(d.items()):
        a = agg.setdefault((k[1], k[2], n // per if per else 0), collections.defaultdict(list))
        for c, x in v.items():
            a[c].append(x)
    for k, a in agg.items():
        print(k, "n=%d" % len(a["_ns"]))
        for c, x in a.items():
            print("   %-22s %14.0f" % (c, sum(x) / len(x)))
