timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_x.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels']: print(k['kernel'], k['launches'], round(k['total_s']*1e3,1),'ms', round(k['tflops'],1),'TF')
PY
