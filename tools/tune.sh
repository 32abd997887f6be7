#!/bin/bash
# sweep the k_bounce scheduling knobs (run on the GPU box)
run() { python bench.py --no-cpu-baseline --no-secondary --seeds-per-launch ${S:-16} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for R in 48 56 62 64; do for T in 1 4 8 16; do CLWH_TUNE_REFILL=$R CLWH_TUNE_STEP=$T run "refill=$R step=$T"; done; done
