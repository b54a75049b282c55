#!/usr/bin/env python3
"""Same-box A/B of the click path's HOST side (round 5): bench.py's latency leg (device-resident / blocking C ABI / reference API p50) in fresh
processes, alternating an environment switch.  usage: python tools/click_host_ab.py IDC_SPIN_SYNC 0 1 [passes]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = ("import sys, json; sys.path.insert(0, %r); import bench; from interactive_deep_colorization_amd import workloads; "
         "print('LAT ' + json.dumps(bench.measure_latency(workloads.random_state_dict(0, 'torch'), 0)))" % REPO)


def main():
    var, values = sys.argv[1], sys.argv[2:4]
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    for p in range(passes):
        for v in values:
            env = dict(os.environ, **{var: v})
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600).stdout
            lat = [json.loads(l[4:]) for l in out.splitlines() if l.startswith("LAT ")]
            if not lat:
                print("%s=%s pass %d: no result" % (var, v, p)); continue
            r = lat[0]
            print("%s=%s pass %d: " % (var, v, p) + "  ".join(
                "%s dev %.4f  c_abi %.4f  api %.4f  api+ab %.4f" % (k, r[k]["device_resident_p50_ms"], r[k]["c_abi_host_call_p50_ms"],
                                                                  r[k].get("api_net_forward_p50_ms") or -1, r[k].get("api_net_forward_then_output_ab_p50_ms") or -1)
                for k in ("fp32", "bf16")), flush=True)


if __name__ == "__main__":
    main()
