#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc results.db: per kernel dispatch, counters summed over instances."""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select dispatch_id, name, counter_name, counter_value, duration from pmc_events"))
d = collections.OrderedDict()
for disp, name, cname, val, dur in rows:
    e = d.setdefault(disp, {"name": name, "dur": dur, "c": collections.defaultdict(float)})
    e["c"][cname] += val
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for disp, e in list(d.items())[skip:]:
    c = e["c"]
    line = "%4d %-44s %9.1f us " % (disp, e["name"][-44:], e["dur"] / 1e3)
    line += " ".join("%s=%.4g" % (k.replace("SQ_", ""), v) for k, v in sorted(c.items()))
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
        cyc = c["SQ_BUSY_CYCLES"] / 32.0           # per shader engine
        line += " | clk=%.2fGHz mfma_util=%.1f%%" % (cyc / e["dur"], 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
        w = c["SQ_WAVE_CYCLES"]
        line += " wait_any=%.0f%% wait_inst=%.0f%% active=%.0f%%" % (100 * c.get("SQ_WAIT_ANY", 0) / w, 100 * c.get("SQ_WAIT_INST_ANY", 0) / w, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w)
    print(line)
