#!/bin/bash
# generic same-box A/B of one compile-time knob of one source file:  tools/ab_knob.sh <file.hip> <KNOB> "<values>" "<sample counts>"
f=$1; k=$2
for v in $3; do
  env $k=$v python physdock_amd/build.py $f > /dev/null 2>&1
  env $k=$v python tools/b20_time.py $4 2>&1 | grep "B=" | sed "s/(PD_ATTN.*//; s/^/$k=$v /"
done
