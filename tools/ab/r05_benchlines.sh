# the three committed bench lines (driver flags, no flags, all-fp16 arrangement)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; O=gpurun_out/final
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_default_steps20.json 2> $O/final_bench20.err
python bench.py > $O/r05_bench_default_noflags.json 2> $O/final_bench.err
python bench.py --steps 300 --warmup 20 --pos-prec fp16 --no-cpu-baseline --no-decode --no-parity > $O/r05_bench_posfp16_steps300.json 2> $O/final_bench_fp16.err
for f in r05_bench_default_steps20 r05_bench_default_noflags r05_bench_posfp16_steps300; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);r=d['roofline']
print('$f', d['value'], r['kernel'][:40], r['frac'], d['config'].get('pos_stream_cus'))"; done
