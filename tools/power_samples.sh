#!/bin/bash
# package power / shader clock samples (rocm-smi, 2 per second) while the bench region runs for ~20 s on the bench operands;
# prints the samples taken under load (> 400 W), the idle reading and the power cap
(python bench.py --steps 5000 --warmup 3 --no-cpu-baseline --no-latency --no-end-to-end --no-peak-probe > /dev/null 2>&1) &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "Package Power\|sclk\|junction" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done | awk '{ if ($0 ~ /[0-9]/) print }' > /tmp/samples.txt
echo "# samples under load (junction C, sclk, package W):"; awk '{ for (i = 1; i <= NF; ++i) if ($i + 0 > 400 && $i !~ /Mhz/) { print; break } }' /tmp/samples.txt | head -40
echo "# idle:"; tail -2 /tmp/samples.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i "Power (W)"
