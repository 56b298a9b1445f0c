#!/bin/bash
# arrangement sweep with the split pair-decomposition position plan: feature sub-batches x position batch multiple
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
export TMPDIR=/tmp
for sb in 2 3 4; do for m in 1 2 3; do
  SLIDE_POS_MULT=$m python bench.py --steps 300 --warmup 20 --sub-batches $sb --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/b_sb${sb}_m$m.json 2> $O/b_sb${sb}_m$m.err
  echo "sub-batches $sb pos_mult $m: $(cut -c80-130 $O/b_sb${sb}_m$m.json)"
done; done
for o in 0 1 2 3; do SLIDE_POS_ORDER=$o python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-decode --no-parity --no-roofline > $O/b_order$o.json 2> $O/b_order$o.err;  echo "order $o: $(cut -c80-130 $O/b_order$o.json)"; done
