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


def traffic_json(d, out_path):
    """HBM-side traffic of the dominant kernel per launch, from the separate FETCH_SIZE and WRITE_SIZE passes
    (KiB -> bytes; raw, see MI355X_MICROARCH.md: the x2 correction applies to wide streams, not to 64-B gathers)."""
    import json

    vals = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        v = [float(r["Counter_Value"]) for r in rows(os.path.join(d, sub, "**", "*counter_collection.csv"))
             if "k_bounce" in r["Kernel_Name"] and r["Counter_Name"] == name]
        if not v:
            return
        vals[name] = sum(v) / len(v) * 1024.0
    with open(out_path, "w") as f:
        json.dump({"kernel": "k_bounce", "fetch_bytes_per_launch": vals["FETCH_SIZE"], "write_bytes_per_launch": vals["WRITE_SIZE"],
                   "launch": "bench.py default: 512^3, 1920x1080, 64 seeds in one launch",
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_gpu.sh"}, f)


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
