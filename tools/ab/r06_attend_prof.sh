cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_attp; rm -rf $O; mkdir -p $O
for v in 0 1; do
SLIDE_MODULE_FUSE_ATTEND=$v SLIDE_MODULE_PREC=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$v -o t -- python tools/time_decode.py > $O/d$v.log 2>&1
rm -f $O/d$v/*kernel_trace.csv $O/d$v/*/*kernel_trace.csv
f=$(find $O/d$v -name "*kernel_stats.csv" | head -1)
echo "== fuse $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%-90s %5s %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
print("total kernel ms %.1f" % (tot/1e6))
PY
done
