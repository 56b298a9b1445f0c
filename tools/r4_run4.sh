cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/gputest2.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/b_full20.json 2>gpurun_out/b_full20.err
cat gpurun_out/gputest2.log; tail -5 gpurun_out/smoke.log; cat gpurun_out/b_full20.json
