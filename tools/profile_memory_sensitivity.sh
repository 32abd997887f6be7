#!/bin/bash
# What does a request cost k_bounce, by the level that serves it?  Run on the GPU box (via gpurun) from the repo root:
#   tools/profile_memory_sensitivity.sh OUT
# 1. tools/probes/gather_levels_probe: independent random 1-byte gathers from tables of 16 KiB .. 8 GiB (L1 / L2 / Infinity Cache / HBM);
#    tools/probes/gather_probe: the same as dependent chains at 8..32 waves per CU.
# 2. Experiment builds of the library (never shipped; built HERE before the call, see below) against the shipped one, alternating:
#    -DCLVR_EXP_STEP_LOADPAD=N  N more loads per march step from the line just requested (L1 hits: address unit / L1 sensitivity)
#    -DCLVR_EXP_STEP_L2PAD=N    N more loads per march step from random lines of the first MiB of the step bytes (L1 misses, L2 hits)
#    -DCLVR_EXP_FAR_SC1=T       the fetch after a step of >= T voxels is an agent-scope load (leaves no line in L1); T = 0: every step fetch
# Build (in the container, before gpurun): for each NAME FLAGS pair
#   hipcc <the flags of cl_volume_renderer_amd/build.py> FLAGS -shared -o tools/ab/libclwhip_NAME.so <csrc sources>
# with NAME in cur (no flag), loadpad1, loadpad3, l2pad1, l2pad2, farsc1_0, farsc1_3, farsc1_8; and the two probes with
#   hipcc -O3 --offload-arch=gfx950 tools/probes/X.hip -o tools/probes/X
out=${1:-gpurun_out/memsens}; mkdir -p "$out"
for p in gather_levels_probe gather_probe; do
  [ -x tools/probes/$p ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/$p.hip -o tools/probes/$p || exit 1
  timeout -k 10 300 ./tools/probes/$p > "$out/$p.txt" 2>&1 || exit 1
done
tools/ab_list.sh 3 cur loadpad1 loadpad3 > "$out/loadpad.txt" 2>&1
tools/ab_list.sh 3 cur l2pad1 l2pad2 > "$out/l2pad.txt" 2>&1
tools/ab_list.sh 3 cur farsc1_0 farsc1_3 farsc1_8 > "$out/farsc1.txt" 2>&1
cat "$out"/*.txt
