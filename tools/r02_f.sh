#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02f
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "click or conv or deconv" > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
cd $R/tools/ablate
{
echo "== conv_click bf16 512@32 ks8"; ./ablate_TIMING 1 32 512 1 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click bf16 512@32 ks4 (2 chunks/WG)"; ./ablate_TIMING 1 32 512 1 1 4 1 4 9 4 | grep -v "block "
echo "== conv_click bf16 512@32 dil2 ks8"; ./ablate_TIMING 1 32 512 2 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click bf16 256@64 ks4"; ./ablate_TIMING 1 64 256 1 1 4 1 4 9 4 | grep -v "block "
echo "== conv_click bf16 128@128 ks2"; ./ablate_TIMING 1 128 128 1 1 4 1 4 9 2 | grep -v "block "
echo "== conv_click fp32 512@32 ks16"; ./ablate_TIMING 1 32 512 1 1 4 0 4 9 16 | grep -v "block "
echo "== conv_click fp32 512@32 ks8"; ./ablate_TIMING 1 32 512 1 1 4 0 4 9 8 | grep -v "block "
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
cd $R
python tools/click_sweep.py --child > $OUT/click_default.json 2>&1; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r02f/click_default.json') if l.startswith('{')][-1])
for p in r: print(p, r[p]['p50_us'], r[p]['sum_layers_us']); print(r[p]['layers_us'])
PY
