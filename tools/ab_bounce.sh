#!/bin/bash
# A/B timing of k_bounce on one box: tools/ab_bounce.sh LIB_A LIB_B [time_bounce.py arguments]; the two builds alternate, three runs each
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  CLWH_LIBRARY=$a python tools/time_bounce.py "$@" 2>&1 | grep k_bounce
  CLWH_LIBRARY=$b python tools/time_bounce.py "$@" 2>&1 | grep k_bounce
done
