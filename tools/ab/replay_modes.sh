#!/bin/bash
# usage (on the GPU box): tools/ab/replay_modes.sh  -- ms/step of the replay modes (graphs / threaded eager) x feature sub-batches
cd $GRAFT_REPO_ROOT
grep -m1 "model name" /proc/cpuinfo; nproc
for rep in 1 2; do
for cfg in "0 128,128" "0 86,85,85" "2 128,128" "2 86,85,85"; do
  set -- $cfg
  echo -n "EAGER=$1 SUB=$2: "
  SLIDE_EAGER=$1 SLIDE_SUB_SIZES=$2 python bench.py --no-cpu-baseline --no-roofline --steps 1000 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
