#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 600 python tools/click_sweep.py > $OUT/click_sweep.txt 2>&1
grep -v "^\[{" $OUT/click_sweep.txt | head -120
