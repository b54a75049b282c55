cd tools/ablate
A=${A:-./ablate_TIMING}
$A 32 32 512 1 4 2 1 1 9 | grep -v "   block"
$A 32 64 256 1 4 2 1 1 9 | grep -v "   block"
$A 32 256 128 1 2 4 1 1 9 | grep -v "   block"
$A 32 128 128 1 2 4 1 1 9 | grep -v "   block"
$A 32 128 128 1 2 4 1 3 | grep -v "   block"
$A 32 64 256 1 4 2 1 3 | grep -v "   block"
