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
    for sub in ("pmc_fetch", "pmc_write", "pmc_req", "pmc_sq"):
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
    """HBM-side traffic per launch of the render kernels from the separate FETCH_SIZE / WRITE_SIZE / TCC request passes
    (counter KiB -> bytes, RAW: no correction applied here; what a counted fetch request is worth in bytes for this
    access pattern is calibrated by tools/probes/miss_bytes_probe.hip)."""
    import json

    def mean_of(sub, kernel, counter):
        v = [float(r["Counter_Value"]) for r in rows(os.path.join(d, sub, "**", "*counter_collection.csv"))
             if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
        return sum(v) / len(v) if v else None

    out = {}
    for kernel in ("k_bounce", "k_primary", "k_sdf_front", "k_repack"):
        fetch, write = mean_of("pmc_fetch", kernel, "FETCH_SIZE"), mean_of("pmc_write", kernel, "WRITE_SIZE")
        if fetch is None or write is None:
            continue
        e = {"fetch_bytes_raw": fetch * 1024.0, "write_bytes": write * 1024.0}
        for c in ("TCC_MISS_sum", "TCC_HIT_sum"):
            e[c] = mean_of("pmc_write", kernel, c)
        for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_REQ_sum", "TCC_READ_sum"):
            e[c] = mean_of("pmc_req", kernel, c)
        out[kernel] = e
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2])
