#!/bin/bash
# Is this box one of those on which the first launch of a kernel after other kernels is 25-35 % slower?
# If so: A/B of the own-code warm-up (IDC_CODE_WARM) and the warm second launch (IDC_DOUBLE_LAUNCH) on it.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/firstuse
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
L={k:(v[0] if isinstance(v,list) else v) for k,v in d['layers'].items()}
r=lambda a,b: L[a]/L[b]
print("%-28s ms %.4f | conv3_2/3_3 %.3f conv5_1/5_2 %.3f conv9_2 %.4f conv8_1 %.4f conv1 %.4f conv2_1 %.4f" % (sys.argv[2], d['ms_per_forward'], r('conv3_2','conv3_3'), r('conv5_1','conv5_2'), L['conv9_2'], L['conv8_1'], L['conv1_1'], L['conv2_1']))
sys.exit(0 if (r('conv5_1','conv5_2') > 1.12 or r('conv3_2','conv3_3') > 1.12) else 3)
PY
}
IDC_CODE_WARM=0 python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/firstuse/w0.json
show gpurun_out/firstuse/w0.json "warm=0"; slow=$?
IDC_CODE_WARM=1 python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/firstuse/w1.json
show gpurun_out/firstuse/w1.json "warm=1"
if [ $slow -eq 0 ]; then
  echo "FIRST-USE BOX"
  for r in 1 2; do
    IDC_CODE_WARM=0 python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/firstuse/w0_$r.json; show gpurun_out/firstuse/w0_$r.json "warm=0 #$r"
    IDC_CODE_WARM=1 python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/firstuse/w1_$r.json; show gpurun_out/firstuse/w1_$r.json "warm=1 #$r"
  done
  IDC_CODE_WARM=0 IDC_DOUBLE_LAUNCH=1 python tools/quick_layers.py . 2>/dev/null | tail -1 > gpurun_out/firstuse/d1.json; show gpurun_out/firstuse/d1.json "warm=0 double-launch"
else
  echo "ordinary box"
fi
