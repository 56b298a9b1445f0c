#!/bin/bash
# usage (on the GPU box): tools/ops_roofline.sh <tag>  -> gpurun_out/<tag>_ops/{trace,fetch,write,manifest.json} + the report
TAG=${1:-r02}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_ops
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/ops_roofline.py --manifest $OUT/manifest.json > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- python tools/ops_roofline.py --manifest $OUT/manifest_f.json > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- python tools/ops_roofline.py --manifest $OUT/manifest_w.json > $OUT/write.log 2>&1
python tools/ops_roofline_report.py $OUT $OUT/${TAG}_ops_roofline.md
# the raw traces are big: keep only the report, the manifest and the logs' tails
rm -rf $OUT/trace $OUT/fetch $OUT/write
