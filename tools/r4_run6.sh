cd /root/repo; mkdir -p gpurun_out
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
( echo "default"; run
for o in pos feat1 feat2 feat pos+feat1; do echo "only $o"; SLIDE_BENCH_ONLY=$o run; done
echo "prio pos low (0) feat high (-1)"; SLIDE_STREAM_PRIO=0,-1 run
echo "prio pos high feat default"; SLIDE_STREAM_PRIO=-1,0 run
echo "default"; run ) > gpurun_out/subsets.log 2>&1
