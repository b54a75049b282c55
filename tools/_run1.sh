set -x
mkdir -p gpurun_out/r6a
# correctness first: layer tests on the deconv + shortcut launches
timeout 900 python -m pytest tests/test_net_gpu.py tests/test_round5_gpu.py -m gpu -x -q -k "bf16 or ragged or config3 or half_workgroups" > gpurun_out/r6a/tests.log 2>&1; tail -5 gpurun_out/r6a/tests.log
cd tools/ablate
for O in 0 1 0 1; do
  echo "== IDC_DS_ORDER=$O"
  IDC_DS_M16=1 IDC_DS_ORDER=$O ./ablate_BASE 32 128 128 1 2 1 1 2 4 64 | tail -1
  IDC_DS_M16=1 IDC_DS_ORDER=$O ./ablate_BASE 32 64 128 1 2 1 1 2 4 128 | tail -1
  IDC_DS_M16=1 IDC_DS_ORDER=$O ./ablate_BASE 32 32 256 1 2 1 1 2 4 256 | tail -1
done > ../../gpurun_out/r6a/harness.txt 2>&1
for O in 0 1; do echo "== TIMING IDC_DS_ORDER=$O"; IDC_DS_M16=1 IDC_DS_ORDER=$O ./ablate_TIMING 32 128 128 1 2 1 1 2 4 64 | grep -v "   block"; done >> ../../gpurun_out/r6a/harness.txt 2>&1
for S in 1000 2500 5000 8000; do for O in 0 1; do echo "== IDC_STAGGER=$S IDC_DS_ORDER=$O"; IDC_STAGGER=$S IDC_DS_M16=1 IDC_DS_ORDER=$O ./ablate_BASE 32 128 128 1 2 1 1 2 4 64 | tail -1; done; done >> ../../gpurun_out/r6a/harness.txt 2>&1
echo "== TIMING stagger 2500 order 1" >> ../../gpurun_out/r6a/harness.txt
IDC_STAGGER=2500 IDC_DS_M16=1 IDC_DS_ORDER=1 ./ablate_TIMING 32 128 128 1 2 1 1 2 4 64 | grep -v "   block" >> ../../gpurun_out/r6a/harness.txt 2>&1
# conv10_2 shape on v2p<2,2,1> with stagger
for S in 0 1500 4000; do echo "== v2p<2,2> conv10_2 shape IDC_STAGGER=$S"; IDC_STAGGER=$S IDC_ABL_V2=1 ./ablate_BASE 32 256 128 1 2 2 1 1 9 | tail -1; done >> ../../gpurun_out/r6a/harness.txt 2>&1
cd ../..
cat gpurun_out/r6a/harness.txt
python tools/option_ab.py ds_order 0 1 > gpurun_out/r6a/ds_order_ab.txt 2>&1; cat gpurun_out/r6a/ds_order_ab.txt
python tools/option_ab.py stagger 0 2500 conv10_1 conv10_2 conv9_1 conv9_2 conv2_2 > gpurun_out/r6a/stagger_ab.txt 2>&1; cat gpurun_out/r6a/stagger_ab.txt
