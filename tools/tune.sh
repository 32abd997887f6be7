#!/bin/bash
# sweeps k_bounce's scheduling knobs on the headline workload; run on the GPU box:  bash tools/tune.sh > gpurun_out/tune.txt
run() { python bench.py --no-cpu-baseline --no-secondary --seeds-per-launch ${S:-64} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
run "default"
for R in 8 16 32; do
  for T in 8 16 24; do
    CLWH_TUNE_REFILL=$R CLWH_TUNE_STEP=$T run "refill=$R step=$T"
  done
done
for B in 1024 1280 1536 2048; do CLWH_TUNE_BLOCKS=$B run "blocks=$B"; done
for G in 4 8 16; do CLWH_TUNE_GROUP=$G run "group=$G"; done
