cd tools/ablate
for A in ./ablate_BASE ./ablate_SETPRIO ./ablate_NOPIN ./ablate_BASE ./ablate_SETPRIO ./ablate_NOPIN; do
$A 32 32 512 1 4 2 1 1 9
$A 32 256 128 1 2 4 1 1 9
done
