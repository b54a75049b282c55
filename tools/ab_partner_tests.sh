#!/bin/bash
# The "nightly" target of the A/B partner kernels (round 6): the default library ships without conv_igemm_v2 / conv_ds_fused / conv1_1_bf16_kernel and without
# environment knobs; this script builds the -DIDC_AB_PARTNERS library into a scratch directory, swaps it in for the run, executes the GPU tests whose partner
# legs are skipped in the default build (plus the whole suite with "all"), and puts the default library back.
#   gpurun --timeout 1500 -- 'bash tools/ab_partner_tests.sh [all] > gpurun_out/ab_partner_tests.log 2>&1'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/interactive_deep_colorization_amd/csrc
T=$(mktemp -d /tmp/idc_ab_XXXX)
make -C $C -j16 ARCH=gfx950 OBJDIR=$T EXTRA=-DIDC_AB_PARTNERS > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $C/libideepcolor_hip.so $T/default.so
cp $T/libideepcolor_hip.so $C/libideepcolor_hip.so
trap 'cp $T/default.so $C/libideepcolor_hip.so' EXIT
cd $R
if [ "$1" = "all" ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|FAILED\|skipped" | tail -8
else
  python -m pytest tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_round5_gpu.py tests/test_net_gpu.py -m gpu -q 2>&1 | grep -a "passed\|failed\|FAILED\|skipped" | tail -8
fi
IDC_MFMA16=0 python -c "
from interactive_deep_colorization_amd import engine, workloads
e = engine.HipColorizer(64, 64, max_batch=8, precision='bf16'); e.load_state_dict(workloads.random_state_dict(0, 'torch'))
import numpy as np; L, ab, m = workloads.random_batch(8, 64, seed=1); engine.set_tile_policy('large'); e.forward(L, ab, m, 0.0)
print('IDC_MFMA16=0 (environment knob, AB build):', sorted(set(r['kernel'] for r in e.layer_table() if r['kernel'].startswith('conv_igemm_v2'))))"
