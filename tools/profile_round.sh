#!/bin/bash
# One gpurun call: default bench line, then rocprofv3 kernel-trace stats and the PMC passes of the same command
# (bench.py --steps 5 --warmup 2, no latency / CPU / end-to-end / probe legs), the N=1 click-path traces, and the
# single-GPU dry run of the N>1 control flow.  Summaries -> gpurun_out/prof_<tag>/.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-latency --no-cpu-baseline --no-end-to-end --no-peak-probe --no-contract-leg"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o x -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o x -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o x -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o x -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o x -- $CMD > $OUT/pmc_grbm.log 2>&1
cd $R
for p in stats pmc_fetch pmc_write pmc_sq pmc_grbm; do
  f=$(find $OUT/$p -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py $f --family conv > $OUT/${p}_summary.txt
  grep -h '"metric"' $OUT/$p.log | tail -1 > $OUT/${p}_benchline.json
done
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) 0 $OUT/pmc_traffic.json $TAG
# click path (N=1) kernel traces
cd /tmp
for p in bf16 fp32; do
  rocprofv3 --kernel-trace --stats -d $OUT/click_$p -o x -- python $R/tools/click_trace.py $p > $OUT/click_$p.log 2>&1
  f=$(find $OUT/click_$p -name "*.db" | head -1)
  [ -n "$f" ] && python $R/tools/click_trace.py --gaps $f > $OUT/click_${p}_trace.txt && python $R/tools/rocpd_summary.py $f --family conv > $OUT/click_${p}_stats.txt
done
cd $R
# N>1 control flow on one GPU: 2 ranks over gloo, both on device 0 (NOT a scaling measurement); round 5: WITHOUT a launcher -- bench.py
# starts its own ranks, exactly what `python3 bench.py --gpus N` does on the driver's 8-GPU node
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --dryrun-single-gpu > $OUT/dryrun_2ranks.json 2> $OUT/dryrun_2ranks.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --dryrun-single-gpu --transport c_abi > $OUT/dryrun_2ranks_c_abi.json 2> $OUT/dryrun_2ranks_c_abi.err
tail -1 $OUT/dryrun_2ranks.json | cut -c1-400
# round 6: the driver's 8-rank job shape on ONE device -- eight processes, two HIP runtimes each, one engine each (small batch: the point is
# the control flow and the queue budget, not the number), and the strong-scaling form
timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 --batch 4 --dryrun-single-gpu --no-latency --no-cpu-baseline > $OUT/dryrun_8ranks.json 2> $OUT/dryrun_8ranks.err
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong --global-batch 16 --dryrun-single-gpu --no-latency --no-cpu-baseline > $OUT/dryrun_2ranks_strong.json 2> $OUT/dryrun_2ranks_strong.err
tail -1 $OUT/dryrun_8ranks.json | cut -c1-300
python tools/extra_configs.py > $OUT/extra_configs.txt 2>&1
python tools/gpu_diag.py 2>/dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $OUT/layer_table.txt
# identical launches on bench / damped / all-zero operands: how much of the forward time is the DVFS response to switching activity
python tools/power_probe.py 30 2>/dev/null | grep "^{" > $OUT/power_probe.txt
# fp32 path (round 3: Winograd F(2x2,3x3) for the 3x3 stride-1 layers): bench line, kernel stats and SQ counters of the same command
python bench.py --precision fp32 --no-cpu-baseline --no-end-to-end --no-peak-probe --no-latency > $OUT/bench_fp32.json 2>/dev/null
python bench.py --precision fp32 --option winograd=0 --no-cpu-baseline --no-end-to-end --no-peak-probe --no-latency > $OUT/bench_fp32_direct.json 2>/dev/null
cd /tmp
CMD32="python $R/bench.py --precision fp32 --steps 3 --warmup 1 --no-latency --no-cpu-baseline --no-end-to-end --no-peak-probe"
rocprofv3 --kernel-trace --stats -d $OUT/stats_fp32 -o x -- $CMD32 > $OUT/stats_fp32.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq_fp32 -o x -- $CMD32 > $OUT/pmc_sq_fp32.log 2>&1
cd $R
for p in stats_fp32 pmc_sq_fp32; do
  f=$(find $OUT/$p -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py $f --family conv > $OUT/${p}_summary.txt
done
# operand-split precisions (round 6): bench lines, kernel stats and SQ counters of the fp16x3 command (the path that carries the 1e-3 contract at the highest rate)
for p in bf16x3 bf16x6 fp16x3 fp16; do
  python bench.py --precision $p --no-cpu-baseline --no-end-to-end --no-peak-probe --no-latency > $OUT/bench_$p.json 2>/dev/null
done
cd /tmp
CMDX3="python $R/bench.py --precision fp16x3 --steps 3 --warmup 1 --no-latency --no-cpu-baseline --no-end-to-end --no-peak-probe"
rocprofv3 --kernel-trace --stats -d $OUT/stats_fp16x3 -o x -- $CMDX3 > $OUT/stats_fp16x3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq_fp16x3 -o x -- $CMDX3 > $OUT/pmc_sq_fp16x3.log 2>&1
cd $R
for p in stats_fp16x3 pmc_sq_fp16x3; do
  f=$(find $OUT/$p -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py $f --family conv > $OUT/${p}_summary.txt
done
# what a pure MFMA loop sustains on this box by operand data (8 s each)
[ -x tools/ubench/mfma_peak ] && tools/ubench/mfma_peak 8 > $OUT/mfma_peak.txt 2>&1
find $OUT -name "*.db" -delete        # keep the summaries, drop the raw databases (size)
ls $OUT
