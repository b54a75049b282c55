#!/bin/bash
# parity record at HEAD (writes gpurun_out/parity_r05_gpu.json) + click latency legs of the bench
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_parity_record_gpu.py -x -q 2>&1 | tail -6
python - <<'PY'
import json
for r in json.load(open('gpurun_out/parity_r05_gpu.json')):
    print("%-58s %-5s %-6s max %.4g mean %.4g q999 %.4g relrms %.3g" % (r['config'][:58], r['precision'], r['weights'], r['max_abs'], r['mean_abs'], r['q999'], r['rel_rms']))
PY
