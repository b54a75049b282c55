#!/usr/bin/env python3
"""bench.py's latency leg alone (BASELINE configs[1]: one 256x256 image, 5 hints): device-resident / C-ABI host call /
whole reference-API net_forward p50, both precisions, as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                                           # noqa: E402
from interactive_deep_colorization_amd import workloads                # noqa: E402

print(json.dumps(bench.measure_latency(workloads.random_state_dict(0, "he"), 0)))
