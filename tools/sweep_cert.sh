#!/bin/bash
# exit certificates (render_kernels.hip, certify_exit) on/off and their threshold, one frame at a time; run on the GPU box:
#   bash tools/sweep_cert.sh [bench.py arguments, e.g. --config 4]
run() { python bench.py --no-cpu-baseline --no-secondary --frames-in-flight 1 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CLWH_TUNE_CERT=%-4s %9.1f Msamples/s  k_bounce %.4f ms' % ('$C', d['value'], d['roofline']['avg_launch_ms']))"; }
for C in 0 4 8 16 32; do CLWH_TUNE_CERT=$C run "$@"; done
