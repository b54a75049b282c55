#!/bin/bash
# anatomy of conv_ds_fused at the conv10_1 / conv8_1-like shapes (in-kernel stamps)
cd tools/ablate
for b in BASE TIMING; do
  timeout 120 ./ablate_$b 32 128 128 1 2 1 1 2 4 64
  timeout 120 ./ablate_$b 32 32 256 1 2 1 1 2 4 256
done 2>&1 | tee ../../gpurun_out/r02_k.txt
