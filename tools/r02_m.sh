#!/bin/bash
# 3-slot early-barrier K loop in conv_igemm_v2 (8-wave tiles) + zero-page halo loads: parity, then timing
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_net_gpu.py tests/test_caffe_branches_gpu.py tests/test_round2_gpu.py -m gpu -x -q 2>&1 | tail -5
cd tools/ablate
for b in BASE TIMING; do
  timeout 120 ./ablate_$b 32 32 512 1 4 2 1 1 9
  timeout 120 ./ablate_$b 32 32 512 2 4 2 1 1 9
  timeout 120 ./ablate_$b 32 128 128 1 2 1 1 2 4 64
done 2>&1 | tee ../../gpurun_out/r02_m.txt
cd ../.. && timeout 300 python bench.py --no-end-to-end --no-peak-probe 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['layers_ms'])"
