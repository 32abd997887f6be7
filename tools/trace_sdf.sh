#!/bin/bash
# tools/trace_sdf.sh TAG -- on the GPU box: kernel trace of tools/time_sdf.py 512, per-kernel totals
set -o pipefail
TAG=${1:-sdf}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/sdftrace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/time_sdf.py 512 > "$OUT/run.txt" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
grep -v amdgpu.ids "$OUT/run.txt"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    d = [x[1] for x in v]
    print("%-64s calls %5d total %10.1f us avg %8.2f us min %7.2f max %7.2f" % (k[:64], len(d), sum(d), sum(d) / len(d), min(d), max(d)))
# the last build of the phantom (first 125 front launches of the last 250... print every 10th layer of the first phantom build)
fr = sorted(per.get("clvr::k_sdf_front(clvr::SdfFrontArgs)", []))
if fr:
    print("front launch durations of the first build, every 8th layer:", " ".join("%.1f" % x[1] for x in fr[:125:8]))
    gaps = [(fr[i + 1][0] - fr[i][0]) / 1e3 for i in range(0, 124)]
    print("start-to-start of consecutive layers (us): mean %.1f min %.1f max %.1f" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
PY
