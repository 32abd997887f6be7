#!/bin/bash
# tools/profile_miss_bytes.sh -- run on the GPU box: the miss-bytes probe un-profiled (timings) and under rocprofv3 --pmc
# (what FETCH_SIZE / TCC_EA0_RDREQ / TCC_MISS report per launch with a KNOWN number of missing lines), counters in
# their own passes.  Output: gpurun_out/miss_bytes/.
set -o pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/miss_bytes
mkdir -p "$OUT"
export TMPDIR=/tmp
P=./tools/probes/miss_bytes_probe
# the probe binary is not tracked (tools/probes/*_probe is git-ignored): build it when it is missing or older than its source
if [ ! -x "$P" ] || [ "$P.hip" -nt "$P" ]; then
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$P.hip" -o "$P" || exit 1
fi
$P > "$OUT/timing.txt" || exit 1
cat "$OUT/timing.txt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/p_fetch" -- $P > /dev/null 2> "$OUT/p_fetch.err" || { tail -5 "$OUT/p_fetch.err"; exit 1; }
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/p_req" -- $P > /dev/null 2> "$OUT/p_req.err" || { tail -5 "$OUT/p_req.err"; exit 1; }
rocprofv3 --pmc TCC_READ_sum TCC_HIT_sum --output-format csv -d "$OUT/p_rd" -- $P > /dev/null 2> "$OUT/p_rd.err" || tail -3 "$OUT/p_rd.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(dict)
for f in glob.glob(out + "/p_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for (k, c), v in per.items():
        acc[k][c] = sorted(v)[-1][1]  # the timed (second) launch of each kernel
with open(out + "/counters.txt", "w") as fh:
    for k in sorted(acc):
        line = "%-60s " % k[:60] + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items()))
        print(line); fh.write(line + "\n")
PY
