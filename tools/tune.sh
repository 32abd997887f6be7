#!/bin/bash
# sweeps k_bounce's scheduling knobs on the headline workload (one frame at a time); run on the GPU box:  bash tools/tune.sh
run() { python bench.py --no-cpu-baseline --no-secondary --frames-in-flight 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s %9.1f Msamples/s  k_bounce %.4f ms' % ('$1', d['value'], d['roofline']['avg_launch_ms']))"; }
run "default"
for R in 8 16 24 32; do
  for T in 8 16 24 32; do
    CLWH_TUNE_REFILL=$R CLWH_TUNE_STEP=$T run "refill=$R step=$T"
  done
done
for B in 1024 1280 1536 2048 3072; do CLWH_TUNE_BLOCKS=$B run "blocks=$B"; done
for K in 2 3 4 5; do CLWH_TUNE_CHUNK_BLOCK_LOG2=$K run "chunk_block_log2=$K"; done
