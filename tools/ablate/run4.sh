cd tools/ablate
A=./ablate_PRE
for n in 1 32; do $A $n 256 128 1 2 4 1 1 9 | grep -v "   block"; done
$A 32 32 512 1 4 2 1 1 9 | grep -v "   block"
$A 1 32 512 1 4 2 1 1 9 | grep -v "   block"
