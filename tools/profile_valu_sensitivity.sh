#!/bin/bash
# Is k_bounce bound by VALU issue?  The same launch with N extra dependent FMAs per event phase / per step iteration (experiment builds
# of the library: -DCLVR_EXP_EVENT_PAD=N, -DCLVR_EXP_STEP_PAD=N): un-profiled launch times, then -- in separate rocprofv3 --pmc passes --
# the issue counters and the L2's memory-side read requests of every build.  usage: tools/profile_valu_sensitivity.sh OUT [time_bounce args]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for n in cur evpad100 evpad300 steppad10; do
  export CLWH_LIBRARY=/root/repo/tools/ab/libclwhip_$n.so
  for r in 1 2; do python3 tools/time_bounce.py "$@" 2>&1 | grep k_bounce; done
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d "$out/$n.sq" -- python3 tools/time_bounce.py --jobs 2 "$@" > /dev/null 2> "$out/$n.sq.err" || { tail -3 "$out/$n.sq.err"; exit 1; }
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/$n.tcc" -- python3 tools/time_bounce.py --jobs 2 "$@" > /dev/null 2> "$out/$n.tcc.err" || { tail -3 "$out/$n.tcc.err"; exit 1; }
  python3 - "$out" $n <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/" + n + ".*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_bounce" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
m = {c: sum(v) / len(v) for c, v in acc.items()}
print("  %s per launch: " % n + "  ".join("%s=%.4g" % (c, m[c]) for c in sorted(m)))
if "SQ_ACTIVE_INST_VALU" in m and "SQ_CYCLES" in m:
    print("  %s: VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs over SQ_CYCLES / 32 = %.1f %%" % (n, 100.0 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["SQ_CYCLES"] / 32)))
PY
done
unset CLWH_LIBRARY
CLWH_LIBRARY=/root/repo/tools/ab/libclwhip_stats.so python3 tools/time_bounce.py --jobs 1 "$@" 2>&1 | grep "bounce stats" | tail -1
