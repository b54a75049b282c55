#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02e
rm -rf $OUT; mkdir -p $OUT
cd $R/tools/ablate
{
for b in TIMING SAMEW NOSTORE BOTH; do echo "== conv_click $b"; ./ablate_$b 1 32 512 1 1 4 1 4 9 8 | grep -v "block "; done
echo "== conv_click N=2"; ./ablate_TIMING 2 32 512 1 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click N=4"; ./ablate_TIMING 4 32 512 1 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click N=1 ksplit 4 (2 chunks: does not fit -> skip)"
echo "== conv_igemm <1,4> ks8 N=4"; ./ablate_TIMING 4 32 512 1 1 4 1 5 9 8 | grep -v "block "
echo "== v2 <4,2> N=1 (32 WGs)"; ./ablate_TIMING 1 32 512 1 4 2 1 1 9 | grep -v "block "
echo "== v2 <4,2> N=8 (256 WGs... 64)"; ./ablate_TIMING 8 32 512 1 4 2 1 1 9 | grep -v "block "
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
