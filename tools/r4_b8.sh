#!/bin/bash
for mm in 1048576 400000 200000; do
echo "wgrad_halo_min_m=$mm $(EVE_WGRAD_HALO_MIN_M=$mm python bench.py --batch 8 --steps 40 --warmup 10 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3))")"
done
for mm in 1048576 400000; do
echo "B=4 wgrad_halo_min_m=$mm $(EVE_WGRAD_HALO_MIN_M=$mm python bench.py --batch 4 --steps 40 --warmup 10 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['ms_per_step'],3))")"
done
