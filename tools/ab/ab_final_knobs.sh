run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
echo default; run
echo TAIL_OCC3; SLIDE_TAIL_OCC3=1 run
echo HEAD_UPDATE; SLIDE_HEAD_UPDATE=1 run
echo GEMM_CHAIN256; SLIDE_GEMM_CHAIN=256 run
echo sb2; run --sub-batches 2
echo default; run
echo graph; run --replay graph
echo PAIR_NORM_V2; SLIDE_PAIR_NORM_V2=1 run
