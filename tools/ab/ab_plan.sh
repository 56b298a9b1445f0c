run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; }
export SLIDE_PAIR_NORM_V2=0 SLIDE_TAIL8=0
echo "BODY=1 sb3"; run
echo "BODY=0 sb3"; SLIDE_BODY=0 run
echo "BODY=0 sb4"; SLIDE_BODY=0 run --sub-batches 4
echo "BODY=0 sb2"; SLIDE_BODY=0 run --sub-batches 2
echo "BODY=0 CHAIN=0 sb3"; SLIDE_BODY=0 SLIDE_SA_CHAIN=0 run
echo "BODY=0 sb3 graph"; SLIDE_BODY=0 run --replay graph
echo "BODY=0 sb3 again"; SLIDE_BODY=0 run
