#!/bin/bash
# round-5 verification set (GPU box): full gpu suite, smoke, ops roofline with the coalesced kNN output
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/gputest.log; cat $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
