#!/bin/bash
# ds_fused S part with the 3-slot ring; A/B of the v2 3-slot loop against the 2-slot loop (same zero-page halo loads)
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_net_gpu.py tests/test_caffe_branches_gpu.py -m gpu -x -q > gpurun_out/r02_n_pytest.txt 2>&1; tail -2 gpurun_out/r02_n_pytest.txt
cd tools/ablate
for rep in 1 2; do for b in BASE RING2; do
  timeout 120 ./ablate_$b 32 32 512 1 4 2 1 1 9
  timeout 120 ./ablate_$b 32 32 512 2 4 2 1 1 9
  timeout 120 ./ablate_$b 32 256 128 1 2 4 1 1 9
done; done 2>&1 | tee ../../gpurun_out/r02_n.txt
for b in BASE TIMING; do
  timeout 120 ./ablate_$b 32 128 128 1 2 1 1 2 4 64
  timeout 120 ./ablate_$b 32 32 256 1 2 1 1 2 4 256
done 2>&1 | tee -a ../../gpurun_out/r02_n.txt
cd ../.. && timeout 300 python bench.py --no-end-to-end --no-peak-probe 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['slowest_layers_ms'])"
