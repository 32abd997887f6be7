#!/bin/bash
# tools/sweep.sh "FLAGS1" "FLAGS2" ... -- on the GPU box: rebuild libclwhip.so with each set of extra hipcc flags and run the
# headline bench once per build (A/B of compile-time experiment switches); restores the default build at the end.
set -o pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/sweep
mkdir -p "$OUT"
i=0
for FLAGS in "$@" ""; do
  i=$((i+1))
  CLVR_EXTRA_HIPCC_FLAGS="$FLAGS" python3 -m cl_volume_renderer_amd.build --force > "$OUT/build$i.log" 2>&1 || { tail -5 "$OUT/build$i.log"; exit 1; }
  python3 bench.py --no-cpu-baseline --no-secondary $BENCH_ARGS > "$OUT/bench$i.json" 2> "$OUT/bench$i.err" || { tail -5 "$OUT/bench$i.err"; exit 1; }
  python3 - "$OUT/bench$i.json" "$FLAGS" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("flags=%-50r value=%9.1f Msamples/s  ms_per_step=%7.4f  k_bounce=%7.4f ms" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"] if "roofline" in d else float("nan")))
PY
  grep "bounce stats" "$OUT/bench$i.err" | tail -2
done
