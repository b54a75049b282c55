#!/bin/bash
# same-box A/B of two builds: _ab/ (a git worktree of the commit to compare against, built in place) vs the tree.
# Boxes differ by up to 10 % in sustained clock, so only numbers from one call are comparable.
for r in 1 2 3; do
  for d in _ab .; do
    (cd $d && python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-end-to-end --no-peak-probe 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$d', j['value'], j['ms_per_step'], j['roofline']['frac'])")
  done
done
