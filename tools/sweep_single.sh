#!/bin/bash
# tools/sweep_single.sh "FLAGS1" "FLAGS2" ... -- like tools/sweep.sh, reporting the single-pass numbers (tools/single_pass.sh)
for FLAGS in "$@" ""; do
  CLVR_EXTRA_HIPCC_FLAGS="$FLAGS" python3 -m cl_volume_renderer_amd.build --force > /dev/null 2>&1 || exit 1
  echo "flags='$FLAGS'"
  bash tools/single_pass.sh A=1
done
