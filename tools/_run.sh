for cfg in "EVE_HALO_PERSIST_ALL=0" "EVE_HALO_PERSIST_ALL=1"; do
  echo "== $cfg"; env $cfg python tools/bench_conv.py 2>&1 | grep -E "l[1234]_3x3"
done > gpurun_out/bench_conv.txt 2>&1
