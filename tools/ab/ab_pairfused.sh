# A/B of the fused per-point GEMM + pair-table launch (SLIDE_PAIR_FUSED) in bench.py's arrangement
run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
echo "fused"; run
echo "two-launch"; SLIDE_PAIR_FUSED=0 run
echo "fused"; run
echo "two-launch"; SLIDE_PAIR_FUSED=0 run
