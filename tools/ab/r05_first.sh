#!/bin/bash
# round-5 first measurement set (GPU box): per-op times of the position plans, the default (pos split) bench line at the
# driver's flags and at 300 steps, the same with the opt-in fp16 position plan, kernel trace of the default arrangement
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a; O=gpurun_out/r05a
export TMPDIR=/tmp
python tools/time_pos_ops.py fp16 split > $O/pos_ops_b256.log 2>&1
B=512 python tools/time_pos_ops.py split > $O/pos_ops_b512.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity > $O/bench_300.json 2> $O/bench_300.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --pos-prec fp16 > $O/bench_300_posfp16.json 2> $O/bench_300_posfp16.err
SLIDE_BENCH_ONLY=feat python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_featonly.json 2> $O/bench_300_featonly.err
SLIDE_BENCH_ONLY=pos python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/bench_300_posonly.json 2> $O/bench_300_posonly.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity --no-decode --replay graph > $O/trace_bench.log 2>&1
rm -f $O/trace/*kernel_trace.csv
timeout 900 python -m pytest tests/test_hip_engine.py -q -m gpu -k "benched or optin or smoke" > $O/gputest_subset.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
for f in bench_steps20 bench_300 bench_300_posfp16 bench_300_featonly bench_300_posonly; do echo $f; cut -c1-260 $O/$f.json; done
tail -3 $O/gputest_subset.log; tail -3 $O/smoke.log; tail -30 $O/pos_ops_b256.log
