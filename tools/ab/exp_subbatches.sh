for q in 4 8; do for sb in 3 4 5 6; do
echo -n "Q=$q sb=$sb: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 300 --warmup 20 --sub-batches $sb --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('sub_batches'))"
done; done
