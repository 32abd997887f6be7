#!/bin/bash
# tools/profile_gpu.sh TAG [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel trace + stats, then
# separate PMC passes (never combined with tracing), all into gpurun_out/prof_TAG/.  Every pass runs the same command:
# bench.py with ONE timed region of 2 frame jobs (plus its 2 warm-up jobs), one frame at a time, no secondary measurements.
# The summaries worth keeping are copied to profiles/ by hand afterwards.
set -o pipefail
TAG=${1:-run}; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-secondary --single-region --frames-in-flight 1 --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- $BENCH > /dev/null 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d "$OUT/pmc_req" -- $BENCH > /dev/null 2> "$OUT/pmc_req.err" || { tail -5 "$OUT/pmc_req.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq" -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err" || { tail -5 "$OUT/pmc_sq.err"; exit 1; }
python3 tools/summarize_prof.py "$OUT" "$OUT/traffic.json" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
