#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_round4_gpu.py -x -q -k deconv 2>&1 | tail -3
bash tools/ab_layers.sh 2>&1 | grep -E "ms/forward|conv8_1|conv9_1|conv10_1|conv10_2|conv5_2|conv1_1"
