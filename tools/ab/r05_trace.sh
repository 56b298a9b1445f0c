# raw kernel trace of a short run of the arrangement (graph replay: the profiler makes eager replay host-bound), for gap / duration analysis
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r05trace; rm -rf $O; mkdir -p $O
for v in ${VARIANTS:-1 0}; do
SLIDE_TAIL_RX=$v rocprofv3 --kernel-trace --output-format csv -d $O/rx$v -o t -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-decode --no-roofline --replay graph > $O/bench_rx$v.log 2>&1
tail -1 $O/bench_rx$v.log | cut -c1-200
done
find $O -name "*agent_info.csv" -delete; du -sh $O
