#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into a small text summary: per-kernel stats from
the kernel trace and per-dispatch means of the PMC counters (FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md's HBM section prescribes for wide streams -- reported raw AND corrected)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def main(d):
    print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
    per = defaultdict(list)
    for r in rows(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in per.values()) or 1.0
    print("%-60s %8s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "max_us", "pct"))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("%-60s %8d %12.1f %12.2f %12.2f %6.1f%%" % (k[:60], len(v), sum(v), sum(v) / len(v), max(v), 100 * sum(v) / tot))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in rows(os.path.join(d, sub, "**", "*counter_collection.csv")):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        print("\n== %s: mean per dispatch ==" % sub)
        for k, cs in acc.items():
            if not any(t in k for t in ("render", "sdf", "bounce", "primary", "repack", "fixup", "resolve")):
                continue
            for c, v in sorted(cs.items()):
                m = sum(v) / len(v)
                extra = ""
                if c == "FETCH_SIZE":
                    extra = "  (KiB raw; x1024 = %.1f MB; gfx950 wide-stream correction x2 = %.1f MB)" % (m * 1024 / 1e6, 2 * m * 1024 / 1e6)
                if c == "WRITE_SIZE":
                    extra = "  (KiB; = %.1f MB)" % (m * 1024 / 1e6)
                print("%-50s %-24s n=%-4d %16.1f%s" % (k[:50], c, len(v), m, extra))


if __name__ == "__main__":
    main(sys.argv[1])
