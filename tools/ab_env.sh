#!/bin/bash
# same-box A/B of this tree under different environment settings: tools/ab_env.sh "VAR=a" "VAR=b" ...
i=0
for e in "$@"; do
  for r in 1 2; do env $e python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/abenv_${i}_$r.json; done
  i=$((i+1))
done
python - "$@" <<'PY'
import json, sys
envs = sys.argv[1:]
runs = [[json.load(open('gpurun_out/abenv_%d_%d.json' % (i, r))) for r in (1, 2)] for i in range(len(envs))]
print("ms/forward: " + " | ".join("%s %s" % (e, [x['ms_per_forward'] for x in rr]) for e, rr in zip(envs, runs)))
for k in runs[0][0]['layers']:
    ms = lambda v: v[0] if isinstance(v, list) else v
    vals = [min(ms(x['layers'].get(k, 0)) for x in rr) for rr in runs]
    print("%-14s " % k + "  ".join("%.4f" % v for v in vals) + "   " + "  ".join("%+5.1f%%" % (100 * (v - vals[0]) / vals[0]) if vals[0] else "" for v in vals[1:]))
PY
