cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_b1; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err ) 2> $O/time.txt
tail -3 $O/time.txt; tail -5 $O/bench20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_b1/bench20.json'))
print(d['value'], d['ms_per_step'], d['config']['replay'], d['config']['replay_auto'], d['config']['host_enqueue_ms_per_step'])
print(json.dumps(d.get('configs'), indent=None)[:3000])
print(json.dumps(d.get('parity'))[:1500])
print(json.dumps(d.get('decode'))[:600])
print(d['roofline']['kernel'], d['roofline']['frac'])
PY
python -m pytest tests/test_bench_contract.py -x -q 2>&1 | tail -15
