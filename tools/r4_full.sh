#!/bin/bash
mkdir -p gpurun_out/full5
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/full5/pytest.log 2>&1
tail -5 gpurun_out/full5/pytest.log
python bench.py --batch 8 --steps 30 --warmup 10 --no-cpu-baseline --no-c3 --no-c5 --no-points > gpurun_out/full5/b8.json 2> gpurun_out/full5/b8.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full5/b8.json'))
print(d['value'], d['ms_per_step'])
ks=d.get('kernels_ms_per_step',{})
print('sum', sum(ks.values()))
for k,v in sorted(ks.items(), key=lambda kv:-kv[1])[:40]: print('%8.4f %s'%(v,k[:110]))
PY
