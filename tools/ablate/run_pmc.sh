# PMC pass over the BASE ablate binary (one kernel shape per call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_abl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/a -o x -- $R/tools/ablate/ablate_BASE 32 32 512 1 4 2 1 1 9 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU -d $OUT/b -o x -- $R/tools/ablate/ablate_BASE 32 32 512 1 4 2 1 1 9 > $OUT/b.log 2>&1
tail -3 $OUT/a.log; tail -3 $OUT/b.log
find $OUT -name "*.db" | while read f; do python $R/tools/rocpd_summary.py $f --skip 2 | sed -n '/PMC/,$p'; done
