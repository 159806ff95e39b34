#!/bin/bash
# lab: per-kernel time of ONE steady-state pass of the conditioning trunk (+ prepare_dit) under rocprofv3 --kernel-trace: the last
# of seven passes (tools/last_pass_stats.py: from the last pd_atom_pair_init on), so first-pass weight packing is left out
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trunk
mkdir -p $OUT
cd /tmp
PD_TRUNK_ONLY=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python $R/tools/trunk_time.py --samples ${1:-64} > $OUT/trace.log 2>&1
python $R/tools/last_pass_stats.py $OUT/trace atom_pair_init_kernel 60
find $OUT -name "*.csv" -size +1M -delete
