cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json 2> gpurun_out/bench_driver_args.err; tail -c 300 gpurun_out/bench_driver_args.json
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/r06f; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/call_trace -o p -- python $R/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > /dev/null 2>&1)
python tools/last_pass_stats.py $OUT/call_trace atom_pair_init_kernel 40 > gpurun_out/r06f_call_b64_steady_state.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
python tools/trunk_time.py --samples 64 2>&1 | grep -i "graph replay" > gpurun_out/trunk_final.txt
bash tools/profile_trunk.sh 64 > gpurun_out/r06f_trunk_kernels.txt 2>&1
bash tools/trunk_pmc.sh > gpurun_out/r06f_trunk_mfma_util.txt 2>&1
cat gpurun_out/trunk_final.txt; head -4 gpurun_out/r06f_call_b64_steady_state.txt
