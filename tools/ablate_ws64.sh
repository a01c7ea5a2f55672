#!/bin/bash
# Timing ablations of conv3x3_ws64_kernel on the GPU box (rebuilds conv_igemm.o per variant; results are WRONG, times only)
for A in "$@"; do
  touch eve_amd/csrc/conv_igemm.hip
  EVE_HIPCC_FLAGS="-DEVE_WS64_ABLATE=$A" python -c "from eve_amd import build as b; b.build(verbose=False)" > /dev/null 2>&1
  echo "ABLATE=$A $(python tools/bench_conv.py 1920 2>/dev/null | grep -E 'l1_3x3')"
done
touch eve_amd/csrc/conv_igemm.hip
python -c "from eve_amd import build as b; b.build(verbose=False)" > /dev/null 2>&1
