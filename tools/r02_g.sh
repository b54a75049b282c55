#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02g
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_ops_gpu.py tests/test_net_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
cd $R/tools/ablate
{
echo "== conv_click bf16 512@32 ks8"; ./ablate_TIMING 1 32 512 1 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click bf16 512@32 ks8 wp=2"; ./ablate_TIMING 1 32 512 1 1 2 1 4 9 8 | grep -v "block "
echo "== conv_click fp32 512@32 ks16"; ./ablate_TIMING 1 32 512 1 1 4 0 4 9 16 | grep -v "block "
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
