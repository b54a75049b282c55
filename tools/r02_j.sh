#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02j
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_net_gpu.py -m gpu -x -q -k "v3 or bf16_within or ragged or config3" > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
cd $R/tools/ablate
{
echo "== v2 <4,2> 512@32 N=32"; ./ablate_TIMING 32 32 512 1 4 2 1 1 9 | grep -v "block "
echo "== v3 512@32 N=32"; ./ablate_TIMING 32 32 512 1 4 2 1 6 9 | grep -v "block "
echo "== v2 <4,2> 512@32 N=32 dil2"; ./ablate_TIMING 32 32 512 2 4 2 1 1 9 | grep -v "block "
echo "== v3 512@32 N=32 dil2"; ./ablate_TIMING 32 32 512 2 4 2 1 6 9 | grep -v "block "
echo "== v2 256@64 N=32"; ./ablate_TIMING 32 64 256 1 4 2 1 1 9 | grep -v "block "
echo "== v3 256@64 N=32"; ./ablate_TIMING 32 64 256 1 4 2 1 6 9 | grep -v "block "
for i in 1 2; do
echo "== BASE v2"; ./ablate_BASE 32 32 512 1 4 2 1 1 9
echo "== BASE v3"; ./ablate_BASE 32 32 512 1 4 2 1 6 9
done
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
cd $R
bash tools/ab_env.sh "IDC_V3=0" "IDC_V3=1" 2>&1 | tail -36
