#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary --seeds-per-launch ${S:-16} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for W in 4 5 6 8; do cp cl_volume_renderer_amd/_variants/libclwhip_w$W.so cl_volume_renderer_amd/libclwhip.so; run "waves=$W"; S=1 run "waves=$W S=1"; done
