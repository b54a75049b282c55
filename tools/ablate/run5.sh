cd tools/ablate
for A in ./ablate_V22 ./ablate_V22P3 ./ablate_V22P1; do
  $A 32 256 128 1 2 2 1 1 9 | grep -v "   block"
  $A 32 128 128 1 2 2 1 3 | grep -v "   block"
  $A 32 128 128 1 2 2 1 1 9 | grep -v "   block"
done
./ablate_V22P3 32 256 128 1 2 4 1 1 9 | grep -v "   block"
