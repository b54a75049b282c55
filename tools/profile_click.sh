#!/bin/bash
# The N = 1 click-path kernel traces alone (rocprofv3 --kernel-trace --stats of tools/click_trace.py, both precisions) + the launch-by-launch
# table and gap analysis; summaries -> gpurun_out/prof_<tag>/.     gpurun --timeout 600 -- 'bash tools/profile_click.sh r04e'
TAG=${1:-r04e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp32; do
  rocprofv3 --kernel-trace --stats -d $OUT/click_$p -o x -- python $R/tools/click_trace.py $p > $OUT/click_$p.log 2>&1
  f=$(find $OUT/click_$p -name "*.db" | head -1)
  [ -n "$f" ] && python $R/tools/click_trace.py --gaps $f > $OUT/click_${p}_trace.txt && python $R/tools/rocpd_summary.py $f --family conv > $OUT/click_${p}_stats.txt
  rm -rf $OUT/click_$p
done
cd $R
