# run-to-run spread of the default bench line: N fresh processes (a bad stream -> hardware-queue assignment shows as a ~0.98 ms step)
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${N:-10}); do echo "run $i: $(python bench.py --gpus 1 --steps ${STEPS:-100} --warmup ${WARM:-10} --no-cpu-baseline --no-decode --no-parity --no-roofline --no-configs 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d.get('dist',{}).get('host_enqueue_ms_per_step'))")"; done
