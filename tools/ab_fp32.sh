#!/bin/bash
# same-box A/B (see ab_bench.sh) of the fp32 batch path and of the click latencies
for r in 1 2; do
  for d in _ab .; do
    (cd $d && python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --no-latency --no-end-to-end --no-peak-probe 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d fp32 N=32', j['value'], j['ms_per_step'])")
    (cd $d && python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-peak-probe 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d click', j['latency']['fp32']['device_resident_p50_ms'], j['latency']['bf16']['device_resident_p50_ms'])")
  done
done
