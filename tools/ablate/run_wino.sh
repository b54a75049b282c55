#!/bin/bash
# fp32: Winograd F(2x2,3x3) kernel forms <TB,CB> vs the direct kernels on the layer shapes of the click path (N=1) and N=32
cd "$(dirname "$0")"
for F in 12 21 22; do
  echo "== IDC_WINO_FORM=$F"
  for cfg in "1 32 512 1" "1 32 512 2" "1 64 256 1" "1 128 128 1" "1 256 128 1" "1 256 64 1" "32 32 512 1" "32 256 128 1"; do
    set -- $cfg
    IDC_WINO_FORM=$F ./ablate_BASE $1 $2 $3 $4 1 4 0 6 9 | tail -1
  done
done
echo "== direct kernels"
./ablate_BASE 1 32 512 1 1 4 0 4 9 16 | tail -1       # conv_click<float>, split-K as the engine picks (+ an 8 us reduction launch in the network)
./ablate_BASE 1 64 256 1 1 4 0 4 9 8 | tail -1
./ablate_BASE 1 128 128 1 1 4 0 4 9 4 | tail -1
./ablate_BASE 1 256 128 1 2 2 0 0 9 | tail -1         # conv_igemm<float,2,2,1> (conv10_2 at batch 1)
./ablate_BASE 1 256 64 1 1 2 0 0 9 | tail -1
./ablate_BASE 32 32 512 1 2 4 0 0 9 | tail -1
./ablate_BASE 32 256 128 1 2 4 0 0 9 | tail -1
