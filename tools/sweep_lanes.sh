#!/bin/bash
# bench.py with 1 / 2 / 3 frame jobs in flight on configs 2, 5 and 4 (one GPU): value, HBM in use, contexts sharing the derived scene
# usage: tools/sweep_lanes.sh OUTDIR
out=${1:-gpurun_out/lanes}
mkdir -p "$out"
for cfg in 2 5 4; do
  for f in 1 2 3; do
    python bench.py --config $cfg --frames-in-flight $f --no-secondary --no-cpu-baseline > "$out/config${cfg}_lanes${f}.json" 2> "$out/config${cfg}_lanes${f}.err" || exit 1
    python - "$out/config${cfg}_lanes${f}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print("config %s lanes %d: %.1f Msamples/s, %.4f ms per job (one at a time %.4f), HBM in use %.2f GiB, derived scene %.2f GiB shared by %d context(s)" % (
    sys.argv[1].split("config")[1][0], c["frames_in_flight"], d["value"], d["ms_per_step"], c["one_frame_at_a_time"]["ms_per_step"],
    c["hbm_in_use_gib"], c["derived_scene"]["bytes"] / 2 ** 30, c["derived_scene"]["contexts_sharing_it"]), flush=True)
PY
  done
done
