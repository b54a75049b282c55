#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c5
timeout 1500 python -m pytest tests/test_round4_gpu.py tests/test_ops_gpu.py tests/test_net_gpu.py tests/test_round3_gpu.py tests/test_parity_record_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4c5/tests.txt
cat gpurun_out/r4c5/tests.txt
python bench.py --no-cpu-baseline > gpurun_out/r4c5/bench.json 2> gpurun_out/r4c5/bench.err
tail -1 gpurun_out/r4c5/bench.json | cut -c1-1500
