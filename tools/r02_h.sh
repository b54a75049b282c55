#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02h
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
cd $R/tools/ablate
{
echo "== conv_click bf16 512@32 ks8"; ./ablate_TIMING 1 32 512 1 1 4 1 4 9 8 | grep -v "block "
echo "== conv_click fp32 512@32 ks16"; ./ablate_TIMING 1 32 512 1 1 4 0 4 9 16 | grep -v "block "
} > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
cd $R
python tools/click_sweep.py --child > $OUT/click_default.json 2>&1; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r02h/click_default.json') if l.startswith('{')][-1])
for p in r: print(p, r[p]['p50_us'], r[p]['sum_layers_us']); print(r[p]['layers_us'])
PY
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline'].get('frac_of_attainable'), j['roofline'].get('attainable_peak')); print(j.get('end_to_end')); print(j.get('latency'))"
tail -3 $OUT/bench.err
