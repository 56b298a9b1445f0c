run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
echo default; run
echo BODY=0; SLIDE_BODY=0 run
echo SA_CHAIN=0; SLIDE_SA_CHAIN=0 run
echo default; run
echo BODY=0; SLIDE_BODY=0 run
echo sb2; run --sub-batches 2
echo graph; run --replay graph
