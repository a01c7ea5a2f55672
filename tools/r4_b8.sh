#!/bin/bash
for mt in 224 120 60; do
echo "min_tiles=$mt $(EVE_CONV_WG8_MIN_TILES=$mt python bench.py --batch 8 --steps 40 --warmup 10 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3))")"
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
