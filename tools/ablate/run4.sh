cd tools/ablate
A=${A:-./ablate_TIMING}
# fused conv10_1: deconv 128->128 on 128x128 sites + conv 64->128 at 256x256
$A 32 128 128 1 2 4 1 2 4 64
$A 32 128 128 1 2 2 1 2 4 64
# fused conv9_1: deconv 256->128?? (harness uses C for both cin and cout): 128ch
$A 32 64 128 1 2 4 1 2 4 128
# plain deconv-like reference: 4 taps x 4 phases not available; plain conv for scale
$A 32 256 128 1 2 4 1 1 9
