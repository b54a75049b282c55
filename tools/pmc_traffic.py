#!/usr/bin/env python3
"""HBM traffic of one forward from two rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE and
WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) -> profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <fetch results.db> <write results.db> <forwards in the run> [out.json]

Units and the gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in
KiB; on this rocprofv3 FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane)
read stream, which is what every load of these kernels is, so read bytes = 2 * FETCH_SIZE * 1024.
WRITE_SIZE was calibrated on a kernel with a known write volume (the round-1 input-pack kernel, and now
conv1_1: N*H*W*64 bf16 = 268,435,456 B at N=32, counter 262,144 KiB): write bytes = WRITE_SIZE * 1024.
"""
import collections
import json
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    tot = collections.OrderedDict()
    calls = collections.defaultdict(set)
    for disp, name, val in con.execute(
            "select dispatch_id, name, counter_value from pmc_events where counter_name = ?", (counter,)):
        tot[name] = tot.get(name, 0.0) + val
        calls[name].add(disp)
    return tot, {k: len(v) for k, v in calls.items()}


def main():
    fetch_db, write_db, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_traffic.json"
    tag = sys.argv[5] if len(sys.argv) > 5 else "?"
    f, fc = per_kernel(fetch_db, "FETCH_SIZE")
    w, wc = per_kernel(write_db, "WRITE_SIZE")
    if forwards <= 0:        # automatic: model1 (conv1_block_fused, else conv1_1) is launched exactly once per forward
        once = [k for k in fc if "conv1_block_fused" in k] or [k for k in fc if "conv1_1" in k]
        forwards = fc[once[0]] if once else 1
    kernels = {}
    rd_total = wr_total = 0.0
    for name in f:
        rd = 2.0 * f[name] * 1024.0
        wr = w.get(name, 0.0) * 1024.0
        rd_total += rd
        wr_total += wr
        kernels[name] = {"launches_per_forward": fc[name] / forwards,
                         "read_bytes_per_launch": rd / fc[name],
                         "write_bytes_per_launch": wr / max(wc.get(name, 1), 1)}
    conv = [k for k in kernels if "conv" in k]
    conv_launches = sum(fc[k] for k in conv)
    conv_bytes = sum(2.0 * f[k] * 1024.0 + w.get(k, 0.0) * 1024.0 for k in conv)
    res = {"tag": tag, "forwards_in_run": forwards,
           "hbm_bytes_per_forward": (rd_total + wr_total) / forwards,
           "read_bytes_per_forward": rd_total / forwards,
           "write_bytes_per_forward": wr_total / forwards,
           "conv_family_bytes_per_launch": conv_bytes / max(conv_launches, 1),
           "conv_family_launches_per_forward": conv_launches / forwards,
           "correction": "read = 2 * FETCH_SIZE KiB (gfx950 wide-read under-count), write = WRITE_SIZE KiB",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
