cd /root/repo; mkdir -p gpurun_out
Q="--steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode"
run() { python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
( echo "default"; run; echo "pos split"; run --pos-prec split; echo "all split"; run --prec split --steps 50; echo "all fp32"; run --prec fp32 --steps 50 ) > gpurun_out/split_arr.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/gputest6.log
