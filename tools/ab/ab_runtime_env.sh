run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-parity --no-roofline --no-decode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'))"; }
echo default; run
for e in AMD_DIRECT_DISPATCH=0 HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=100 HSA_NO_SCRATCH_RECLAIM=1 HIP_FORCE_QUEUE_PROFILING=0 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_DYNAMIC_QUEUES=0; do echo $e; env $e bash -c "$(declare -f run); run"; done
echo default; run
