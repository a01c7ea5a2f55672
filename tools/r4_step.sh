#!/bin/bash
# quick A/B on the GPU box:  bash tools/r4_step.sh "<pytest -k expr or empty>" [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4step
rm -rf $O; mkdir -p $O
cd $R
if [ -n "$1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -15 > $O/pytest.log
fi
python tools/bench_in.py > $O/bench_in.txt 2>&1
python bench.py --no-cpu-baseline --no-c3 --no-points ${@:2} > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest.log; cat $O/bench_in.txt | head -12; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'])
for k,v in sorted(d.get('kernels_ms_per_step',{}).items(), key=lambda kv:-kv[1])[:24]: print('%8.4f %s'%(v,k))
print(d.get('kernel_groups_ms_per_step'))
PY
