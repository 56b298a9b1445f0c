# one-GPU batch curve of the default arrangement: per-256-shape step time vs per-GPU batch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_curve; mkdir -p $O
for b in 256 512 1024 2048; do
  python bench.py --gpus 1 --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/arr_$b.json 2> $O/arr_$b.err
  echo "arrangement batch $b: $(python -c "import json;d=json.load(open('$O/arr_$b.json'));print(d['value'], d['ms_per_step'], d['ms_per_step']*256/$b, d['config']['sub_batches'], d['config']['host_enqueue_ms_per_step'])")"
done
# feature chains alone / one feature chain alone / position chain alone
for only in feat feat1 pos; do for b in 256 1024; do
  SLIDE_BENCH_ONLY=$only python bench.py --gpus 1 --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/${only}_$b.json 2> $O/${only}_$b.err
  echo "$only batch $b: $(python -c "import json;d=json.load(open('$O/${only}_$b.json'));print(d['ms_per_step'], d['ms_per_step']*256/$b, d['config']['sub_batches'])")"
done; done
