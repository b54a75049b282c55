#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c3
timeout 600 python -m pytest tests/test_round4_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4c3/tests.txt
cat gpurun_out/r4c3/tests.txt
cd tools/ablate
for M in 1 2 1 2; do
  echo "== IDC_DS_M16=$M"
  IDC_DS_M16=$M ./ablate_BASE 32 128 128 1 2 1 1 2 4 64 | tail -1
  IDC_DS_M16=$M ./ablate_BASE 32 64 128 1 2 1 1 2 4 128 | tail -1
  IDC_DS_M16=$M ./ablate_BASE 32 32 256 1 2 1 1 2 4 256 | tail -1
done > ../../gpurun_out/r4c3/ablate_ds.txt 2>&1
for M in 1 2; do echo "== TIMING IDC_DS_M16=$M"; IDC_DS_M16=$M ./ablate_TIMING 32 128 128 1 2 1 1 2 4 64 | grep -v "   block"; done >> ../../gpurun_out/r4c3/ablate_ds.txt 2>&1
cat ../../gpurun_out/r4c3/ablate_ds.txt
cd ../..
bash tools/ab_env.sh "IDC_DS_M16=1" "IDC_DS_M16=2" > gpurun_out/r4c3/ab_ds.txt 2>&1
cat gpurun_out/r4c3/ab_ds.txt
