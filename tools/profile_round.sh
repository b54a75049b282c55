#!/bin/bash
# One gpurun call: default bench line, then rocprofv3 kernel-trace stats and the two HBM PMC passes of the
# same command (bench.py --steps 5 --warmup 2, no latency/CPU legs).  Summaries -> gpurun_out/prof_<tag>/.
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r01b'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-latency --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o x -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o x -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o x -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o x -- $CMD > $OUT/pmc_sq.log 2>&1
cd $R
for p in stats pmc_fetch pmc_write pmc_sq; do
  f=$(find $OUT/$p -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py $f --family conv > $OUT/${p}_summary.txt
  grep -h '"metric"' $OUT/$p.log | tail -1 > $OUT/${p}_benchline.json
done
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) 12 $OUT/pmc_traffic.json
find $OUT -name "*.db" -delete        # keep the summaries, drop the raw databases (size)
ls -la $OUT
