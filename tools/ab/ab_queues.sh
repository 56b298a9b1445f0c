# hardware-queue count (GPU_MAX_HW_QUEUES) x feature sub-batch count in bench.py's arrangement
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('launches_per_step'), d['config'].get('host_enqueue_ms_per_step'))"; }
for q in 1 2 3 4 6; do
for sb in 3 4; do
echo "q$q sb$sb"; GPU_MAX_HW_QUEUES=$q run --sub-batches $sb
done; done
