#!/usr/bin/env python
"""bench.py's cpu_baseline at several thread counts (the line reports 32 of the box's 256 hardware threads).
One warm-up + two timed oracle train steps (B = 2 clips x T = 30, 128 x 128) per thread count, each under a wall-clock budget;
the log is kept in profiles/rNN_cpu_baseline_threads.log.      python tools/cpu_baseline_threads.py [budget seconds per count]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from oracle import sequence  # noqa: E402
from oracle.config import OracleConfig  # noqa: E402
from oracle.eye_net import EyeNet as OracleEyeNet  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
print('host cpus: %d' % (os.cpu_count() or 1))
for threads in (8, 16, 32, 64, 128, os.cpu_count() or 1):
    torch.set_num_threads(threads)
    cfg = OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001)
    torch.manual_seed(0)
    net = OracleEyeNet(cfg)
    opt = sequence.make_optimizer(net.parameters(), cfg)
    batch = bench.synthetic_eyenet_batch(2, 30, 128, 'cpu', 123)
    t0 = time.perf_counter()
    sequence.eyenet_train_step(net, opt, batch, cfg)
    warm = time.perf_counter() - t0
    done, t0 = 0, time.perf_counter()
    while done < 2 and (time.perf_counter() - t0) + warm < budget:
        sequence.eyenet_train_step(net, opt, batch, cfg)
        done += 1
    dt = (time.perf_counter() - t0) / done if done else warm
    print('%3d threads: %.2f s per step (%s) -> %.1f frames/s' % (threads, dt, '%d timed' % done if done else 'warm-up only', 60 / dt), flush=True)
