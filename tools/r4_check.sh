cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/final_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_fp16_steps20.json 2> gpurun_out/final_bench20.err
cat gpurun_out/final_gputest.log; tail -3 gpurun_out/final_smoke.log; cut -c1-400 gpurun_out/r04_bench_fp16_steps20.json
