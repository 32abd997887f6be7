#!/bin/bash
# tools/sweep_single_pass.sh -- on the GPU box: k_bounce's register budget (waves per SIMD) against the single-pass latency of the
# reference's own call pattern (bench.py's reference_exact_mode / drop_in_path) and the headline value; restores the default build.
for W in 4 5 6 8; do
  CLVR_EXTRA_HIPCC_FLAGS="-DCLVR_BOUNCE_WAVES_PER_SIMD=$W" python3 -m cl_volume_renderer_amd.build --force > /dev/null 2>&1
  python3 bench.py --no-cpu-baseline > gpurun_out/single_w$W.json 2> /dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/single_w$W.json')); r=d['reference_exact_mode']; p=d['drop_in_path']
print('waves/SIMD $W: headline', d['value'], ' single-pass k_bounce', r['k_bounce_ms_per_pass'], 'ms; per pass', r['ms_per_pass'], '; render_frame still', p['render_frame_still_camera']['ms_per_frame'], 'device x1', p['render_frame_device_1_pass_per_call']['ms_per_pass'], 'x8', p['render_frame_device_8_passes_per_call']['ms_per_pass'])"
done
python3 -m cl_volume_renderer_amd.build --force > /dev/null 2>&1
