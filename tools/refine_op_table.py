#!/usr/bin/env python
"""Every convolution / InstanceNorm launch of one C3 train step (eager), timed one by one with events: which layer runs which
kernel at what algorithmic bandwidth.  refine_op_table.py [clips] [frames]"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench  # noqa: E402
from eve_amd.kernels import default_kernels  # noqa: E402

ROWS = []


def wrap(k, name, describe):
    inner = getattr(k, name)

    def timed(*a, **kw):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = inner(*a, **kw)
        e1.record()
        torch.cuda.synchronize()
        if out is None and name.endswith('_fused'):
            return out
        ms = e0.elapsed_time(e1)
        desc, mb = describe(a, kw, out)
        ROWS.append((name, desc, k.lib.eve_last_kernel().decode(), ms, mb))
        return out
    setattr(k, name, timed)


def nbytes(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None) / 1e6


def main():
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seq = int(sys.argv[2]) if len(sys.argv) > 2 else 30

    class A(object):
        no_graph = True
    tr, batch, cfg = bench.pipeline_setup(A(), torch.device('cuda:0'), 'c3', clips, seq, 128, 'bf16', False)
    np.random.seed(0)
    for _ in range(2):
        tr.step(batch)
    k = default_kernels()
    sh = lambda t: 'x'.join(str(s) for s in t.shape)
    wrap(k, 'conv2d_fwd', lambda a, kw, o: ('%s w %s s%d acc=%d' % (sh(a[0]), sh(a[1]), a[3], kw.get('accumulate_into') is not None),
                                            nbytes(a[0], o, kw.get('accumulate_into'))))
    wrap(k, 'conv2d_dgrad', lambda a, kw, o: ('%s w %s s%d acc=%d' % (sh(a[0]), sh(a[1]), a[3], kw.get('accumulate_into') is not None),
                                              nbytes(a[0], o, kw.get('accumulate_into'))))
    wrap(k, 'conv2d_wgrad', lambda a, kw, o: ('%s dy %s k%d' % (sh(a[0]), sh(a[1]), a[2]), nbytes(a[0], a[1])))
    wrap(k, 'instnorm_stats', lambda a, kw, o: (sh(a[0]), nbytes(a[0])))
    wrap(k, 'instnorm_act_fwd', lambda a, kw, o: (sh(a[0]), nbytes(a[0], o)))
    wrap(k, 'instnorm_act_bwd', lambda a, kw, o: (sh(a[0]) + (' y' if a[1] is not None else ''), nbytes(a[0], a[1], a[2], o[0], o[1])))
    wrap(k, 'instnorm_fwd_fused', lambda a, kw, o: (sh(a[0]), nbytes(a[0], a[3], o[0])))
    wrap(k, 'instnorm_bwd_fused', lambda a, kw, o: (sh(a[0]) + (' y' if a[1] is not None else ''), nbytes(a[0], a[1], a[2], o[0], o[1])))
    if hasattr(k, 'instnorm_act2_fwd'):
        wrap(k, 'instnorm_act2_fwd', lambda a, kw, o: ('+'.join(sh(t) for t in a[0]), nbytes(*a[0]) + nbytes(*[t for t in o if t is not None])))
    if hasattr(k, 'instnorm_act2_bwd'):
        # (dy_a, dy_b, xs, ...): two passes over the gradients and the sources, one write of the sources' gradients
        wrap(k, 'instnorm_act2_bwd', lambda a, kw, o: ('+'.join(sh(t) for t in a[2]) if isinstance(a[2], (list, tuple)) else sh(a[0]),
                                                        2 * nbytes(a[0], a[1]) + 3 * (nbytes(*a[2]) if isinstance(a[2], (list, tuple)) else 0.0)))
    tr.step(batch)
    agg = {}
    for name, desc, kern, ms, mb in ROWS:
        key = (name, desc, kern[:60])
        c = agg.setdefault(key, [0, 0.0, mb])
        c[0] += 1
        c[1] += ms
    tot = sum(v[1] for v in agg.values())
    print('total timed ms', tot)
    for (name, desc, kern), (n, ms, mb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
        print('%7.3f ms  x%-2d %6.3f each  %5.2f TB/s  %-18s %-40s %s' % (ms, n, ms / n, mb / (ms / n) / 1e3 if mb else 0.0, name, desc, kern))


if __name__ == '__main__':
    main()
