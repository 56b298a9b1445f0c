cd /root/repo; mkdir -p gpurun_out
Q="--no-cpu-baseline --no-parity --no-roofline --no-decode"
for n in 20 20 40 100; do
SLIDE_BENCH_CHAIN_ENDS=1 python bench.py --steps $n --warmup 5 $Q 2>&1 | grep -E "chain ends|metric" | cut -c1-260
done > gpurun_out/chain_ends.log 2>&1
