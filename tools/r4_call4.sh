#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c4
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_ops_gpu.py tests/test_round3_gpu.py -x -q -k "not winograd" 2>&1 | tail -15 > gpurun_out/r4c4/tests.txt
cat gpurun_out/r4c4/tests.txt
bash tools/ab_env.sh "IDC_V2P=0" "IDC_V2P=1" > gpurun_out/r4c4/ab_v2p.txt 2>&1
cat gpurun_out/r4c4/ab_v2p.txt
