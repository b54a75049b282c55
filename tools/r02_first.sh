#!/bin/bash
# round-2 first GPU call: MFMA-peak ubench (data-dependent clock), baseline bench line, N=1 click-path kernel traces
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02a
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 120 tools/ubench/mfma_peak 8 > $OUT/mfma_peak.txt 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp32; do
  rocprofv3 --kernel-trace --stats -d $OUT/click_$p -o x -- python $R/tools/click_trace.py $p > $OUT/click_$p.log 2>&1
  f=$(find $OUT/click_$p -name "*.db" | head -1)
  [ -n "$f" ] && python $R/tools/click_trace.py --gaps $f > $OUT/click_${p}_gaps.txt && python $R/tools/rocpd_summary.py $f --family conv > $OUT/click_${p}_stats.txt
done
find $OUT -name "*.db" -delete
cat $OUT/mfma_peak.txt
head -8 $OUT/click_bf16_gaps.txt; head -8 $OUT/click_fp32_gaps.txt
