#!/bin/bash
# Register / scratch usage of every kernel of the library (hipcc remarks), one line each, per source file.
cd "$(dirname "$0")/../interactive_deep_colorization_amd/csrc"
for f in idc_igemm idc_v2 idc_conv1 idc_heads idc_colour idc_v2m idc_dsm idc_kw idc_wino idc_session; do
  echo "# $f.hip"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include "$@" -c $f.hip -o /tmp/idc_k.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
    awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {s=$NF} /VGPRs Spill/ {sp=$NF} /TotalSGPRs:/ {sg=$NF}
         /LDS Size/ {printf "%-62s vgpr %3s agpr %3s sgpr %3s scratch %4s spill %3s\n", name, v, a, sg, s, sp}' | c++filt | sort -u
done
