#!/bin/bash
# click-path anatomy: in-kernel stamps of conv_click vs conv_igemm (N=1, 512->512 @32x32, bf16) + kernel trace with click on
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02d
rm -rf $OUT; mkdir -p $OUT
cd $R/tools/ablate
{
echo "== conv_click wp=4 ksplit 8 (1 chunk per WG)"; ./ablate_TIMING 1 32 512 1 1 4 1 4 9 8
echo "== conv_click wp=4 ksplit 8, dilated"; ./ablate_TIMING 1 32 512 2 1 4 1 4 9 8
echo "== conv_igemm <2,1> ksplit 4 (current default)"; ./ablate_TIMING 1 32 512 1 2 1 1 5 9 4
echo "== conv_igemm <1,4> ksplit 8"; ./ablate_TIMING 1 32 512 1 1 4 1 5 9 8
echo "== conv_click 256ch @64"; ./ablate_TIMING 1 64 256 1 1 4 1 4 9 4
echo "== conv_igemm 256ch @64 <2,2> ksplit 2"; ./ablate_TIMING 1 64 256 1 2 2 1 5 9 2
echo "== conv_click fp32 512 wp=4 ksplit 16"; ./ablate_TIMING 1 32 512 1 1 4 0 4 9 16
echo "== conv_igemm fp32 <2,1> ksplit 8"; ./ablate_TIMING 1 32 512 1 2 1 0 5 9 8
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/click_on -o x -- python $R/tools/click_trace.py bf16 > $OUT/click_on.log 2>&1
f=$(find $OUT/click_on -name "*.db" | head -1)
[ -n "$f" ] && python $R/tools/click_trace.py --gaps $f > $OUT/click_on_gaps.txt
find $OUT -name "*.db" -delete
sed -n 1,60p $OUT/click_on_gaps.txt
