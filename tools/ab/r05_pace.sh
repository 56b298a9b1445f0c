cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
for rep in 1 2; do for lag in off 0 1 3 8; do
if [ $lag = off ]; then E=""; else E="SLIDE_PACE_LAG=$lag"; fi
env $E SLIDE_BENCH_CHAIN_ENDS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/s20_$lag_$rep.json 2> $O/s20_$lag_$rep.err
echo "steps20 lag=$lag rep $rep: $(python -c "import json;d=json.load(open('$O/s20_$lag_$rep.json'));print(d['value'], d['ms_per_step'])") $(grep 'chain ends' $O/s20_$lag_$rep.err | cut -c60-140)"
done; done
for lag in off 0 1 3 8; do
if [ $lag = off ]; then E=""; else E="SLIDE_PACE_LAG=$lag"; fi
env $E python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/s300_$lag.json 2> $O/s300_$lag.err
echo "steps300 lag=$lag: $(python -c "import json;d=json.load(open('$O/s300_$lag.json'));print(d['value'], d['ms_per_step'])")"
done
