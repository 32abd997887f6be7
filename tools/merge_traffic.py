"""python tools/merge_traffic.py "KEY" gpurun_out/prof_TAG/traffic.json SUMMARY_NAME -- file one profiled configuration's per-kernel
traffic (tools/profile_gpu.sh -> tools/summarize_prof.py) under bench.py's workload key in profiles/r03_traffic.json (round 2:
r02_traffic.json), which bench.py reads for `roofline.traffic`; SUMMARY_NAME is the committed rocprofv3 summary the numbers come from."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, src, summary = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(ROOT, "profiles", "r03_traffic.json")
table = json.load(open(dst)) if os.path.exists(dst) else {}
note = ("GB per launch = TCC_EA0_RDREQ x 128 B + WRITE_SIZE: every L2-side read request on gfx950 is a 128-byte line "
        "(TCC_EA0_RDREQ_32B = 0; FETCH_SIZE tallies it at 64 B), calibrated for random 1-byte gathers by "
        "tools/probes/miss_bytes_probe.hip (profiles/r02_miss_bytes_probe.txt); rocprofv3 --pmc, separate passes, profiles/" + summary)
entry = {}
for kernel, e in json.load(open(src)).items():
    entry[kernel] = {
        "gb_per_launch": (e["TCC_EA0_RDREQ_sum"] * 128.0 + e["write_bytes"]) / 1e9,
        "read_requests_128B": e["TCC_EA0_RDREQ_sum"],
        "fetch_size_raw_bytes": e["fetch_bytes_raw"],
        "write_bytes": e["write_bytes"],
        "tcc_miss": e["TCC_MISS_sum"], "tcc_hit": e["TCC_HIT_sum"], "tcc_req": e["TCC_REQ_sum"],
        "note": note,
    }
table[key] = entry
json.dump(table, open(dst, "w"), indent=1)
print(key, "k_bounce GB per launch %.3f" % entry["k_bounce"]["gb_per_launch"])
