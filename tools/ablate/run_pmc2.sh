cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_abl2
rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; shift; 
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/${tag}_a -o x -- $R/tools/ablate/ablate_BASE "$@" > $OUT/${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/${tag}_b -o x -- $R/tools/ablate/ablate_BASE "$@" > $OUT/${tag}_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM -d $OUT/${tag}_c -o x -- $R/tools/ablate/ablate_BASE "$@" > $OUT/${tag}_c.log 2>&1
for x in a b c; do grep -h "TFLOP" $OUT/${tag}_$x.log; find $OUT/${tag}_$x -name "*.db" | while read f; do python $R/tools/rocpd_summary.py $f --skip 2 | sed -n '/PMC/,$p' | grep -v "^#" | grep -A9 "conv_igemm"; done; done; }
echo "=== fused"; run fused 32 128 128 1 2 4 1 2 4 64
echo "=== plain conv 128ch 256^2"; run plain 32 256 128 1 2 4 1 1 9
