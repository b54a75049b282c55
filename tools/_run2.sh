cd tools/ablate
for N in 1 2 4 8 32; do echo "== N=$N order 0"; IDC_DS_M16=2 IDC_DS_ORDER=0 ./ablate_TIMING $N 128 128 1 2 1 1 2 4 64 | grep -v "   block"; done
for N in 2 32; do echo "== N=$N order 1"; IDC_DS_M16=2 IDC_DS_ORDER=1 ./ablate_TIMING $N 128 128 1 2 1 1 2 4 64 | grep -v "   block"; done
echo "== N=32 order 0 stagger 8000, block starts"; IDC_STAGGER=8000 IDC_DS_M16=2 IDC_DS_ORDER=0 ./ablate_TIMING 32 128 128 1 2 1 1 2 4 64 | tail -8
