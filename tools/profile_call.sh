#!/bin/bash
# lab: per-kernel time of ONE steady-state sample_diffusion call (graph replay) of bench.py at a given sample count: the last of
# four calls under rocprofv3 --kernel-trace (tools/last_pass_stats.py: from the last pd_atom_pair_init on).  usage: profile_call.sh <samples>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}
OUT=$R/gpurun_out/call_b$B
mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python $R/bench.py --samples $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
python $R/tools/last_pass_stats.py $OUT/trace atom_pair_init_kernel 50
find $OUT -name "*.csv" -size +1M -delete
