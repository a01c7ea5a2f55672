#!/bin/bash
mkdir -p gpurun_out/legs
python bench.py --workload c3 --no-cpu-baseline > gpurun_out/legs/bench_c3.json 2> gpurun_out/legs/err.log
python bench.py --workload c5 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/legs/bench_c5.json 2>> gpurun_out/legs/err.log
python - <<'PY'
import json
for f in ('bench_c3','bench_c5'):
    d=json.loads([l for l in open('gpurun_out/legs/%s.json'%f) if l.startswith('{')][0])
    print(f, d['value'], d['ms_per_step'])
    for k in ('roofline','roofline_other_bound'):
        r=d.get(k); print(' ',k, r and {a:r[a] for a in ('kernel','bound','achieved','frac','traffic','launches_per_step','avg_launch_ms','algorithmic_mb_per_launch')})
PY
