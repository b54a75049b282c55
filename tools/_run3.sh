cd tools/ablate
for r in 1 2; do for B in BASE PRODMA NT; do
  echo "== $B"
  IDC_DS_M16=1 IDC_DS_ORDER=0 ./ablate_$B 32 128 128 1 2 1 1 2 4 64 | tail -1
  IDC_DS_M16=1 IDC_DS_ORDER=0 ./ablate_$B 32 64 128 1 2 1 1 2 4 128 | tail -1
  IDC_DS_M16=1 IDC_DS_ORDER=0 ./ablate_$B 32 32 256 1 2 1 1 2 4 256 | tail -1
done; done
for B in TIMING PRODMAT; do echo "== $B"; IDC_DS_M16=1 IDC_DS_ORDER=0 ./ablate_$B 32 128 128 1 2 1 1 2 4 64 | grep -v "   block"; done
