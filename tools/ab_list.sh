#!/bin/bash
# time k_bounce for a list of builds (tools/ab/libclwhip_NAME.so), the whole list REPS times: tools/ab_list.sh REPS NAME [NAME ...] [-- time_bounce.py arguments]
reps=$1; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq $reps); do
  for n in "${names[@]}"; do
    CLWH_LIBRARY=/root/repo/tools/ab/libclwhip_$n.so python tools/time_bounce.py "$@" 2>&1 | grep k_bounce
  done
done
