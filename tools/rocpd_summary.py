#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd SQLite, what `rocprofv3 --kernel-trace --stats` writes on
this image) as text: the per-kernel stats table, the same split by launch geometry, and -- when the
run was a `--pmc` pass -- the counters summed per kernel.

    python tools/rocpd_summary.py <results.db> [--skip N] [--family conv_igemm]

--skip N   ignore the first N dispatches (warm-up / weight upload)
--family   also print one aggregate row for every kernel whose name contains this substring
"""
import argparse
import collections
import sqlite3


def short(name):
    name = name.replace("void ", "").replace("idc::", "")
    return name if len(name) <= 64 else name[:61] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--family", default="conv_igemm")
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    rows = list(con.execute(
        "select dispatch_id, name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count "
        "from kernels order by start"))
    rows = rows[args.skip:]
    if not rows:
        print("no kernel dispatches")
        return
    total = float(sum(r[2] for r in rows))
    print("# kernel stats (%d dispatches, %.3f ms of kernel time)" % (len(rows), total / 1e6))
    print("%-66s %6s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r[1], []).append(r[2])
    for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("%-66s %6d %11.1f %10.2f %10.2f %10.2f %6.2f" % (short(name), len(d), sum(d) / 1e3, sum(d) / len(d) / 1e3,
                                                            min(d) / 1e3, max(d) / 1e3, 100.0 * sum(d) / total))
    fam = [r[2] for r in rows if args.family in r[1]]
    if fam:
        print("%-66s %6d %11.1f %10.2f %10.2f %10.2f %6.2f" % ("[family *%s*]" % args.family, len(fam), sum(fam) / 1e3,
                                                            sum(fam) / len(fam) / 1e3, min(fam) / 1e3, max(fam) / 1e3,
                                                            100.0 * sum(fam) / total))
    print()
    print("# by launch geometry (kernel, workgroups, threads, LDS bytes, arch VGPR + AGPR, SGPR)")
    print("%-50s %8s %5s %7s %9s %5s %6s %10s" % ("kernel", "wgs", "thr", "lds", "vgpr+agpr", "sgpr", "calls", "avg_us"))
    geo = collections.OrderedDict()
    for r in rows:
        key = (r[1], r[3] // max(r[4], 1), r[4], r[5], r[6], r[7], r[8])
        geo.setdefault(key, []).append(r[2])
    for key, d in geo.items():
        print("%-50s %8d %5d %7d %5d+%-3d %5d %6d %10.2f" % (short(key[0])[:50], key[1], key[2], key[3], key[4], key[5],
                                                           key[6], len(d), sum(d) / len(d) / 1e3))
    try:
        pmc = list(con.execute("select dispatch_id, name, counter_name, counter_value from pmc_events"))
    except sqlite3.Error:
        pmc = []
    if pmc:
        keep = set(r[0] for r in rows)
        agg = collections.OrderedDict()
        ndisp = collections.defaultdict(set)
        for disp, name, cname, val in pmc:
            if disp not in keep:
                continue
            agg.setdefault(name, collections.defaultdict(float))[cname] += val
            ndisp[name].add(disp)
        print()
        print("# PMC counters, summed over instances/XCDs, averaged per dispatch")
        for name, c in agg.items():
            n = len(ndisp[name])
            print("%-66s calls %d" % (short(name), n))
            for k in sorted(c):
                print("    %-32s %.6g" % (k, c[k] / n))


if __name__ == "__main__":
    main()
