timeout 1200 python tools/train_sanity.py 160 > gpurun_out/train_sanity.log 2>&1
