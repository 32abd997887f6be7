#!/bin/bash
# tools/profile_pmc_extra.sh TAG -- extra PMC passes for k_bounce (issue mix); run on the GPU box.
# (A pass with TCP_* / TA_* counters aborted inside rocprofv3 on this pool and is not part of the script.)
set -o pipefail
TAG=${1:-extra}; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-secondary --single-region --frames-in-flight 1 --steps 2 --warmup 1 $*"
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- $BENCH > /dev/null 2> "$OUT/p$i.err" || { tail -5 "$OUT/p$i.err"; exit 1; }
done <<'SETS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_bounce" not in r["Kernel_Name"]:
            continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c in sorted(acc):
    v = acc[c]
    print("%-44s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
