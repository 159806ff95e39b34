export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/trunkprof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trunkprof -o t -- python $R/tools/trunk_only.py cfg1 > $R/gpurun_out/trunkprof/run.log 2>&1 )
grep trunk gpurun_out/trunkprof/run.log
DB=$(find gpurun_out/trunkprof -name "*.db" | head -1)
python tools/prof_summary.py $DB > gpurun_out/trunkprof/kernel_stats.txt 2>&1; head -40 gpurun_out/trunkprof/kernel_stats.txt | cut -c1-175
find gpurun_out/trunkprof -name "*.db" -delete
