#!/bin/bash
# round-5 knob sweep of the FEATURE plan's opt-in variants inside the new default arrangement (experiments library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
run() { tag=$1; shift; env SLIDE_EXPERIMENTS=1 "$@" python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/$tag.json 2> $O/$tag.err; echo "$tag: $(python -c "import json;d=json.load(open('$O/$tag.json'));print(d['value'], d['ms_per_step'], d['config']['launches_per_step'])" 2>/dev/null || tail -1 $O/$tag.err)"; }
run base0 A=1
run tail_occ3 SLIDE_TAIL_OCC3=1
run gx_n64_off SLIDE_GX_N64=0
run gx_n64w_off SLIDE_GX_N64W=0
run body_off SLIDE_BODY=0
run sa_chain_off SLIDE_SA_CHAIN=0
run chain_p_off SLIDE_CHAIN_P=0
run head_update SLIDE_HEAD_UPDATE=1
run gemm_chain256 SLIDE_GEMM_CHAIN=256
run pair_norm_v2 SLIDE_PAIR_NORM_V2=1
run tail8 SLIDE_TAIL8=1
run gx_dual_off SLIDE_GX_DUAL=0
run merge_q_off SLIDE_MERGE_Q=0
run tail_wide8 SLIDE_TAIL_WIDE=8
run base1 A=1
run graph_replay A=1 SLIDE_REPLAY=graph
run threads_replay A=1 SLIDE_REPLAY=threads
