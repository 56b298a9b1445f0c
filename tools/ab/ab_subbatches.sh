run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
echo sb3; run
echo sb2; run --sub-batches 2
echo sb4; run --sub-batches 4
echo "sb3 graph"; run --replay graph
echo "sb3 threads"; run --replay threads
