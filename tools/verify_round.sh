# round-4 verification set (GPU box): full gpu suite, smoke, the driver's bench line, the no-flags bench line, profile refresh
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_fp16_steps20.json 2> gpurun_out/final_bench20.err
python bench.py > gpurun_out/r04_bench_fp16_noflags.json 2> gpurun_out/final_bench.err
bash tools/ops_roofline.sh r04b > gpurun_out/ops_r04b.log 2>&1
cat gpurun_out/final_gputest.log; tail -3 gpurun_out/final_smoke.log; cut -c1-400 gpurun_out/r04_bench_fp16_steps20.json; cut -c1-200 gpurun_out/r04_bench_fp16_noflags.json
