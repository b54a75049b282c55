#!/bin/bash
# Builds the standalone timing harness of the conv kernels (tuning tool, not part of the library):
#   ablate_BASE    plain timing          ablate_TIMING  + in-kernel cycle stamps (IDC_TIMING)
# usage of the binaries: see ablate.hip (N HW C halo wm wp prec v2 ntaps [C2]); run1..6.sh / run_pmc*.sh are the
# gpurun scripts used during round 1.
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DABL_NAME=\"BASE\" ablate.hip -o ablate_BASE
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DABL_NAME=\"TIMING\" -DIDC_TIMING ablate.hip -o ablate_TIMING
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DABL_NAME=\"PROBE\" -DIDC_TIMING -DIDC_STEP_PROBE ablate.hip -o ablate_PROBE
ls -la ablate_BASE ablate_TIMING ablate_PROBE
