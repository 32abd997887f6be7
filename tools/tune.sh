#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary --seeds-per-launch ${S:-64} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for U in 1 2 4 8 16 32 64; do CLWH_TUNE_SEEDS_PER_UNIT=$U run "seeds_per_unit=$U"; done
