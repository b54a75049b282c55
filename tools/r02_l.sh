#!/bin/bash
# conv_ds_fused with the barrier-free D part: parity first, then timing
timeout 900 python -m pytest tests/test_net_gpu.py tests/test_caffe_branches_gpu.py -m gpu -x -q 2>&1 | tail -5
cd tools/ablate
for b in BASE TIMING; do
  timeout 120 ./ablate_$b 32 128 128 1 2 1 1 2 4 64
  timeout 120 ./ablate_$b 32 32 256 1 2 1 1 2 4 256
done 2>&1 | tee ../../gpurun_out/r02_l.txt
cd ../.. && timeout 300 python bench.py --no-end-to-end --no-peak-probe 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('slowest_layers_ms'))"
