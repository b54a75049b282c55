#!/usr/bin/env python3
"""Effective shader clock per kernel from a rocpd_summary of a `--pmc GRBM_GUI_ACTIVE` pass:
clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / average kernel duration (MI355X_MICROARCH.md, 'DVFS give-back').
Usage: python tools/clocks_from_grbm.py <pmc_grbm_summary.txt> [tag]"""
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
avg = {}
for ln in txt:
    m = re.match(r"^(\S.*?\))\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
    if m:
        avg.setdefault(m.group(1), float(m.group(4)))
print("# effective shader clock per kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, both from the same")
print("# rocprofv3 --pmc GRBM_GUI_ACTIVE pass of `bench.py --steps 5 --warmup 2` (profiles/%s_pmc_grbm_summary.txt)." % tag)
print("# For comparison (profiles/r02_mfma_peak.txt): a pure v_mfma_f32_32x32x16_bf16 loop sustains 2.39 GHz on zero operands")
print("# and 1.78-1.81 GHz on uniform random operands.")
name = None
for ln in txt:
    m = re.match(r"^(\S.*?)\s+calls (\d+)\s*$", ln)
    if m:
        name = m.group(1)
        continue
    m = re.match(r"^\s+GRBM_GUI_ACTIVE\s+([\d.e+]+)", ln)
    if m and name in avg and "conv" in name:
        print("%-40s avg %8.2f us  -> %.3f GHz" % (name, avg[name], float(m.group(1)) / 8 / avg[name] / 1e3))
