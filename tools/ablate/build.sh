#!/bin/bash
# builds every ablation variant of the conv kernel (tuning harness)
cd "$(dirname "$0")"
build() { name=$1; shift; hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DABL_NAME=\"$name\" "$@" ablate.hip -o ablate_$name 2>&1 | grep -E "error|Error"; }
build BASE &
build NO_WLOAD -DIDC_ABL_NO_WLOAD &
build NO_WWRITE -DIDC_ABL_NO_WWRITE &
build NO_BARRIER -DIDC_ABL_NO_BARRIER &
build NO_MFMA -DIDC_ABL_NO_MFMA &
build MFMA_DSREAD_ONLY -DIDC_ABL_NO_WLOAD -DIDC_ABL_NO_WWRITE -DIDC_ABL_NO_BARRIER &
build MFMA_ONLY -DIDC_ABL_NO_WLOAD -DIDC_ABL_NO_WWRITE -DIDC_ABL_NO_BARRIER -DIDC_ABL_NO_DSREAD &
build NO_DSREAD -DIDC_ABL_NO_DSREAD &
wait
ls -la ablate_* 
