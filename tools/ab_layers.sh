#!/bin/bash
# same-box A/B: per-layer times of _ab/ (worktree of the commit to compare against, built in place) vs this tree
for r in 1 2; do python tools/quick_layers.py _ab 2>/dev/null | tail -1 > gpurun_out/ab_old_$r.json; python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/ab_new_$r.json; done
python - <<'PY'
import json
o=[json.load(open('gpurun_out/ab_old_%d.json'%r)) for r in (1,2)]; n=[json.load(open('gpurun_out/ab_new_%d.json'%r)) for r in (1,2)]
print("ms/forward old %s new %s" % ([x['ms_per_forward'] for x in o], [x['ms_per_forward'] for x in n]))
for k in o[0]['layers']:
    ms=lambda v: v[0] if isinstance(v,list) else v
    a=min(ms(x['layers'][k]) for x in o); b=min(ms(x['layers'].get(k,0)) for x in n)
    print("%-14s %.4f -> %.4f  %+5.1f%%" % (k, a, b, 100*(b-a)/a if a else 0))
PY
