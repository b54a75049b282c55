#!/bin/bash
# conv_ds_fused: persistent tile walk (default) vs one tile per workgroup, the three deconv + shortcut shapes of the N=32 forward
cd "$(dirname "$0")"
for P in 1 0 1 0; do
  echo "== IDC_DS_PERSIST=$P"
  IDC_DS_PERSIST=$P ./ablate_BASE 32 128 128 1 2 1 1 2 4 64 | tail -1      # conv10_1: deconv 128->128 @128 + short 64->128 @256
  IDC_DS_PERSIST=$P ./ablate_BASE 32 64 128 1 2 1 1 2 4 128 | tail -1      # ~conv9_1 (harness takes Cin == Cout: 128)
  IDC_DS_PERSIST=$P ./ablate_BASE 32 32 256 1 2 1 1 2 4 256 | tail -1      # ~conv8_1 (256)
done
IDC_DS_PERSIST=1 ./ablate_TIMING 32 128 128 1 2 1 1 2 4 64 | grep -v "   block"
