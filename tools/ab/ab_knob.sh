# A/B of one plan knob in bench.py's arrangement: tools/ab/ab_knob.sh SLIDE_GEMM_CHAIN=0
run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
for r in 1 2; do
echo "default"; run
echo "$@"; env "$@" bash -c "$(declare -f run); run"
done
