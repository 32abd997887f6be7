#!/bin/bash
# k_bounce (one ray per lane) against k_bounce2 (two rays per lane, CLWH_TUNE_BOUNCE_RAYS=2) on one box: launch times in alternation,
# the scheduling statistics of a -DCLVR_BOUNCE_STATS build, then the issue and L2 counters in separate rocprofv3 --pmc passes.
# usage: tools/profile_two_rays.sh OUT [time_bounce.py arguments]
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
L=/root/repo/tools/ab
for rep in 1 2 3; do
  for rays in 1 2; do
    echo -n "rays $rays: "; CLWH_TUNE_BOUNCE_RAYS=$rays CLWH_LIBRARY=$L/libclwhip_cur.so python3 tools/time_bounce.py "$@" 2>&1 | grep k_bounce
  done
done
for step in 16 24 32 40; do for refill in 8 16 32; do
  echo -n "rays 2, step threshold $step, refill threshold $refill: "; CLWH_TUNE_STEP=$step CLWH_TUNE_REFILL=$refill CLWH_TUNE_BOUNCE_RAYS=2 CLWH_LIBRARY=$L/libclwhip_cur.so python3 tools/time_bounce.py --jobs 10 "$@" 2>&1 | grep k_bounce
done; done
for rays in 1 2; do
  echo "rays $rays:"; CLWH_TUNE_BOUNCE_RAYS=$rays CLWH_LIBRARY=$L/libclwhip_stats.so python3 tools/time_bounce.py --jobs 1 "$@" 2>&1 | grep "bounce stats" | tail -2
  export CLWH_TUNE_BOUNCE_RAYS=$rays CLWH_LIBRARY=$L/libclwhip_cur.so
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d "$out/rays$rays.sq" -- python3 tools/time_bounce.py --jobs 2 "$@" > /dev/null 2> "$out/rays$rays.sq.err" || { tail -3 "$out/rays$rays.sq.err"; exit 1; }
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/rays$rays.tcc" -- python3 tools/time_bounce.py --jobs 2 "$@" > /dev/null 2> "$out/rays$rays.tcc.err" || { tail -3 "$out/rays$rays.tcc.err"; exit 1; }
  unset CLWH_TUNE_BOUNCE_RAYS CLWH_LIBRARY
  python3 - "$out" rays$rays <<'PY'
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
    print("  %s: VALU busy = %.1f %%" % (n, 100.0 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["SQ_CYCLES"] / 32)))
PY
done
