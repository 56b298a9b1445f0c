cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
for rep in 1 2 3; do for m in 1 2; do
SLIDE_POS_MULT=$m SLIDE_BENCH_CHAIN_ENDS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/s20_m${m}_$rep.json 2> $O/s20_m${m}_$rep.err
echo "mult $m rep $rep: $(cut -c80-125 $O/s20_m${m}_$rep.json) | $(grep 'chain ends' $O/s20_m${m}_$rep.err)"
done; done
