#!/bin/bash
# the single-pass numbers of bench.py (reference-exact mode and the drop-in path) for the current build; run on the GPU box.
# Arguments: environment assignments to try, e.g.  bash tools/single_pass.sh CLWH_TUNE_CERT=0 CLWH_TUNE_CERT=16
for E in "$@"; do
  env $E python3 bench.py --no-cpu-baseline > gpurun_out/single.json 2> /dev/null
  python3 -c "
import json; d=json.loads(open('gpurun_out/single.json').read().strip().split('\n')[-1]); r=d['reference_exact_mode']; p=d['drop_in_path']
print('$E: headline', d['value'], ' single-pass k_bounce', r['k_bounce_ms_per_pass'], 'ms; per pass', r['ms_per_pass'], '; render_frame still', p['render_frame_still_camera']['ms_per_frame'], 'moving', p['render_frame_moving_camera']['ms_per_frame'], 'device x1', p['render_frame_device_1_pass_per_call']['ms_per_pass'], 'x8', p['render_frame_device_8_passes_per_call']['ms_per_pass'])"
done
