#!/usr/bin/env python
"""bench.py -- train-step frames/s of the EVE hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W)

One step = one optimiser step over one batch of synthetic clips already resident in HBM:
NCHW->NHWC/bf16 conversion, EyeNet forward for all T frames and both eyes, the masked losses,
backward, [gradient all-reduce], global-norm clip and Adam.  The workload is BASELINE.json configs[1]
("EyeNet training on 1xMI355X, bf16, B=32 clips of T=30 synthetic frames"), weak scaling: every rank
processes its own B=32 clips.  frames/s = world * B * T / max-over-ranks step time.

Printed JSON (rank 0, one line) also carries
  roofline     -- the dominant kernel group, its algorithmic FLOP/s from HIP events recorded around
                  every launch in the timed region, against the dense bf16 MFMA peak;
  cpu_baseline -- the CPU oracle (plain-torch restatement of the reference, oracle/) running the same
                  train step on this host's cores on a bounded sample (B=2 clips), reported only.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}     # /opt/skills/guides/MI355X_MICROARCH.md:41-42 (dense)
HBM_PEAK_GBS = 8000.0
TORCH_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}
# algorithmic FLOP per frame (one time step of one clip = 2 eye patches), SURVEY.md 8(d) / BASELINE.md 2
EYENET_TRAIN_GFLOP_PER_FRAME_128 = 6.955
# threads of the CPU baseline: the FASTEST count on the 256-thread host of the GPU boxes (profiles/r05_cpu_baseline_threads.log:
# 8 threads 54 frames/s, 16 threads 63, 32 threads 25, 64 threads 10, 128 threads 2, 256 threads: no step within 10 minutes --
# the sample is N = 2 images per op, more threads only add synchronisation).  The line states `cores` and `host_cpus`.
CPU_BASELINE_THREADS = 16


def synthetic_eyenet_batch(B, T, size, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    b = {}
    for side in ('left', 'right'):
        b[side + '_eye_patch'] = (torch.rand((B, T, 3, size, size), generator=g) * 2 - 1)
        b[side + '_h'] = 0.1 * torch.randn((B, T, 2), generator=g)
        b[side + '_g_tobii'] = 0.2 * torch.randn((B, T, 2), generator=g)
        b[side + '_p'] = 2 + 3 * torch.rand((B, T), generator=g)
        b[side + '_g_tobii_validity'] = torch.ones((B, T), dtype=torch.bool)
        b[side + '_p_validity'] = torch.ones((B, T), dtype=torch.bool)
    return {k: v.to(device) for k, v in b.items()}


def cpu_baseline(T, size, steps=3, budget_s=45.0):
    """The oracle's train step on the host cores (bounded sample: B=2 clips of T frames, per-time-step loop
    exactly like the reference).  Thread count: CPU_BASELINE_THREADS, the fastest on the GPU boxes' host (`cores` in the line says so,
    `host_cpus` what the box has)."""
    from oracle import sequence
    from oracle.config import OracleConfig
    from oracle.eye_net import EyeNet as OracleEyeNet
    cores = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(cores)
    cfg = OracleConfig(batch_size=16, weight_decay=0.005, base_learning_rate=0.001)
    torch.manual_seed(0)
    net = OracleEyeNet(cfg)
    opt = sequence.make_optimizer(net.parameters(), cfg)
    B = 2
    batch = synthetic_eyenet_batch(B, T, size, 'cpu', 123)
    t0 = time.perf_counter()
    sequence.eyenet_train_step(net, opt, batch, cfg)           # warm-up
    warm = time.perf_counter() - t0
    done, t0 = 0, time.perf_counter()
    while done < steps and (time.perf_counter() - t0) + warm * (done + 1) / max(done, 1) < budget_s:
        sequence.eyenet_train_step(net, opt, batch, cfg)
        done += 1
    dt = (time.perf_counter() - t0) / done if done else warm
    return {'value': B * T / dt, 'unit': 'frames/s', 'cores': cores, 'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': 'oracle (plain-torch fp32 restatement of the reference, per-time-step loop) EyeNet train '
                      'step, B=%d clips x T=%d, %dx%d, %d timed step(s) after 1 warm-up'
                      % (B, T, size, size, done if done else 0)}


def cpu_baseline_c3(T, steps=3, budget_s=60.0):
    """The oracle's configs[2] train step on the host cores: EVE pipeline (EyeNet frozen forward, RefineNet / CGRU trained,
    geometry, heat-maps, soft-argmax, the 31 losses) + clip + Adam, B=2 clips x T frames, per-frame loop like the reference."""
    from oracle import detweights, eve as oracle_eve
    from oracle.config import OracleConfig
    from oracle.eye_net import EyeNet as OracleEyeNet
    from oracle.refine_net import RefineNet as OracleRefineNet
    cores = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(cores)
    cfg = OracleConfig(os.path.join(HERE, 'configs', 'refine_net.json'), refine_net_rnn_type='CGRU', eye_net_load_pretrained=False)
    eye, ref = detweights.fill_module(OracleEyeNet(cfg), 0), detweights.fill_module(OracleRefineNet(cfg), 1)
    for q in eye.parameters():
        q.requires_grad_(False)
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    B = 2
    batch = detweights.eve_batch(B, T, seed=3)

    def step():
        opt.zero_grad()
        out, _, _ = oracle_eve.eve_forward(eye, ref, dict(batch), cfg, True)
        out['full_loss'].backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), cfg.gradient_clip_amount)
        opt.step()
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    done, t0 = 0, time.perf_counter()
    while done < steps and (time.perf_counter() - t0) + warm < budget_s:
        step()
        done += 1
    dt = (time.perf_counter() - t0) / done if done else warm
    return {'value': B * T / dt, 'unit': 'frames/s', 'cores': cores, 'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': 'oracle EVE train step of configs[2] (EyeNet frozen fwd, RefineNet/CGRU trained, per-frame loop), '
                      'B=%d clips x T=%d, %d timed step(s) after 1 warm-up' % (B, T, done)}


def _profiled_shape(command):
    """(batch, seq, size, dtype) of the workload a profiles/ summary was taken on, from the command line stamped into it (the
    defaults of bench.py / tools/bench_eve.py where a flag is absent)."""
    words = (command or '').split()
    get = lambda flag, dflt: words[words.index(flag) + 1] if flag in words and words.index(flag) + 1 < len(words) else dflt
    c5 = get('--workload', 'c2') == 'c5'
    return (int(get('--batch', 8 if c5 else 32)), int(get('--seq', 120 if c5 else 30)), int(get('--size', 256 if c5 else 128)),
            get('--dtype', 'fp16' if c5 else 'bf16'))


def pmc_traffic(symbol, suffix='_pmc_hbm_per_kernel.json', shape=None):
    """HBM bytes per launch of `symbol` from a committed rocprofv3 --pmc summary (two separate passes, FETCH_SIZE and
    WRITE_SIZE, of this same workload; FETCH doubled as MI355X_MICROARCH.md prescribes for wide reads).  Counters cannot be
    collected from inside the timed process, so this is a measured profile from profiles/ -- but ONLY one taken on exactly
    these kernel sources (tools/pmc_hbm_summary.py stamps eve_amd.build.kernel_tree_sha() into the file) AND on the same
    (batch, seq, size, dtype) as the point that quotes it (`shape`): a symbol's bytes per launch scale with the image count, so
    the B = 32 profile says nothing about a B = 8 launch of the same symbol.  Anything else gives None."""
    from eve_amd.build import kernel_tree_sha
    here = os.path.dirname(os.path.abspath(__file__))
    sha = kernel_tree_sha()
    cands = sorted(f for f in os.listdir(os.path.join(here, 'profiles')) if f.endswith(suffix) and
                   all((tag in f) == (tag in suffix) for tag in ('_c3_', '_c5_'))) \
        if os.path.isdir(os.path.join(here, 'profiles')) else []
    for name in reversed(cands):
        try:
            table = json.load(open(os.path.join(here, 'profiles', name)))
        except (OSError, ValueError):
            continue
        if (table.get('_meta') or {}).get('kernel_tree_sha') != sha:
            continue
        if shape is not None and _profiled_shape((table.get('_meta') or {}).get('command')) != tuple(shape):
            continue
        for k, v in table.items():
            if k.replace('void ', '').strip() == 'eve::' + symbol and v.get('fetch_mb_avg_x2') is not None:
                return (v['fetch_mb_avg_x2'] + (v.get('write_mb_avg') or 0.0)) * 1024 * 1024, 'profiles/' + name
    return None, None


def kernel_roofline(sym, d, mfma_peak, profile_steps, overhead, pmc_suffix=None, shape=None):
    """Roofline of one kernel symbol from its HIP-event launch durations.  The bound is the roof the kernel's ALGORITHMIC
    intensity puts it under: FLOP per byte (operands read once, result written once) against the ridge mfma_peak / 8 TB/s
    (312 FLOP/B in bf16: ResNet layer 1's 3x3 convolutions sit at 288, layers 2-4 at 575 .. 2 300) -- SURVEY.md 8(d)."""
    ms = d['ms'] * 1e-3
    tf = d['flops'] / ms / 1e12
    gbs = d['bytes'] / ms / 1e9
    ai = d['flops'] / d['bytes'] if d['bytes'] else float('inf')
    hbm = d['flops'] == 0 or ai < mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    traffic, traffic_src = pmc_traffic(sym, suffix=pmc_suffix, shape=shape) if pmc_suffix else pmc_traffic(sym, shape=shape)
    r = {'bound': 'hbm' if hbm else 'mfma', 'kernel': 'eve::' + sym,
         'achieved': gbs if hbm else tf, 'peak': HBM_PEAK_GBS if hbm else mfma_peak, 'unit': 'GB/s' if hbm else 'TFLOP/s',
         'frac': (gbs / HBM_PEAK_GBS) if hbm else (tf / mfma_peak), 'traffic': traffic,
         'traffic_unit': 'bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_src,
         'launches_per_step': d['launches'] / profile_steps, 'avg_launch_ms': d['ms'] / d['launches'],
         'algorithmic_gflop_per_launch': d['flops'] / d['launches'] / 1e9,
         'algorithmic_mb_per_launch': d['bytes'] / d['launches'] / 1e6, 'flop_per_byte': ai,
         'mfma_frac': tf / mfma_peak, 'hbm_frac': gbs / HBM_PEAK_GBS,
         'event_pair_overhead_ms_subtracted': overhead}
    return r


# algorithmic GFLOP per frame of the pipeline workloads, SURVEY.md 8(d): RefineNet(CGRU) train 9.619; EyeNet forward 2.370 /
# train 6.955 at 128 x 128, conv part x 4 at 256 x 256 (9.476 / 27.8)
PIPELINE_GFLOP = {('c3', 128): 2.370 + 9.619, ('c5', 256): 27.8 + 9.619, ('c5', 128): 6.955 + 9.619, ('c3', 256): 9.476 + 9.619}


def pipeline_setup(args, device, which, batch_clips, seq, size, dtype_name, use_graph, distributed=False, seed=1, rnn='CGRU'):
    """The eve_amd.EVE train step as a (trainer, batch) pair.
    which = 'c3': BASELINE configs[2] -- configs/refine_net.json with refine_net_rnn_type=CGRU: EyeNet frozen and
      forward-only, offset augmentation, gaze geometry, heat-maps, RefineNet trained (fused conv-GRU scan), soft-argmax, the 31
      losses / metrics, clip, Adam;
    which = 'c5': BASELINE configs[4] -- the same pipeline with BOTH networks trained (EyeNet's angular + pupil losses switched
      on), T = 120 frames of 256 x 256 patches in float16."""
    import eve_amd
    from eve_amd import synthetic, train
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(HERE, 'configs', 'refine_net.json'))
    cfg.import_dict({'refine_net_rnn_type': rnn, 'eye_net_load_pretrained': False})
    if which == 'c5':
        cfg.import_dict({'eye_net_frozen': False, 'loss_coeff_g_ang_initial': 1.0, 'loss_coeff_pupil_size': 1.0})
    model = eve_amd.EVE()
    dt = TORCH_DTYPE[dtype_name]
    model.eye_net.compute_dtype = model.refine_net.compute_dtype = dt
    synthetic.fill_module(model.eye_net, seed=0)
    synthetic.fill_module(model.refine_net, seed=1)
    model = model.to(device).train()
    tr = train.eve_trainer(model, cfg, distributed=distributed, use_graph=use_graph)
    tr.static_inputs = 'alias'
    small = synthetic.eve_batch(4, seq, seed=seed)
    if size != 128:
        g = torch.Generator().manual_seed(4000 + seed)
        for side in ('left', 'right'):
            small[side + '_eye_patch'] = torch.rand((4, seq, 3, size, size), generator=g) * 2 - 1
    reps = (batch_clips + 3) // 4
    batch = {kk: torch.cat([v] * reps, dim=0)[:batch_clips].contiguous().to(device) for kk, v in small.items()}
    return tr, batch, cfg


def bench_pipeline(args, device, k, which, batch_clips, seq, size, dtype_name, steps, warmup, rnn='CGRU', profile=True):
    """One pipeline operating point on this GPU -> the `c3` / `c5` object of the JSON line: ms/step, frames/s, the HBM roofline
    of its dominant kernel (RefineNet is HBM-bound by construction).  rnn = 'CLSTM': the cell the reference's
    shipped src/configs/refine_net.json:55 names (SURVEY 8(d) quotes configs[2] on the CGRU override; `c3_clstm` is the point that
    shows the shipped configuration is not a slow path)."""
    import numpy as np
    import eve_amd
    use_graph = not args.no_graph
    tr, batch, cfg = pipeline_setup(args, device, which, batch_clips, seq, size, dtype_name, use_graph, rnn=rnn)
    np.random.seed(0)
    for _ in range(max(2, warmup)):
        terms = tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        terms = tr.step(batch)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    label = {'c3': 'BASELINE configs[2]: configs/refine_net.json + CGRU through eve_amd.EVE (EyeNet frozen fwd, RefineNet trained, '
                   'geometry / heat-maps / soft-argmax / 31 losses, clip, Adam)',
             'c5': 'BASELINE configs[4]: long-sequence stress through eve_amd.EVE, EyeNet AND RefineNet/CGRU trained (refine_net.json '
                   'with the EyeNet losses on), hidden state of all T frames on-chip'}[which]
    gf = PIPELINE_GFLOP[(which, size)]        # (CGRU figure; the CLSTM cell's one 128 -> 256 convolution is within 0.1 % of it)
    if rnn != 'CGRU':
        label = label.replace('CGRU', rnn)
    out = {'workload': '%s, B=%d x T=%d, %dx%d patches, %s' % (label, batch_clips, seq, size, size, dtype_name),
           'ms_per_step': ms, 'value': batch_clips * seq / (ms * 1e-3), 'unit': 'frames/s', 'steps': steps, 'hip_graph': use_graph,
           'final_loss': float(terms['full_loss'].detach()), 'optimizer': tr.optimizer_state(),
           'step_algorithmic_tflops': gf * batch_clips * seq / (ms * 1e-3) / 1e3}
    out['step_mfma_frac'] = out['step_algorithmic_tflops'] / MFMA_PEAK_TFLOPS[dtype_name]
    if profile and not args.no_roofline:
        tr._eager_step(batch)
        torch.cuda.synchronize()
        k.start_profile()
        for _ in range(args.profile_steps):
            tr._eager_step(batch)
        prof = k.stop_profile()
        by_kernel = prof.pop('_by_kernel', {})
        overhead = prof.pop('_event_overhead_ms', None)
        # the heaviest kernel symbol that sits UNDER the HBM roof by its algorithmic intensity (RefineNet's outer levels, the
        # InstanceNorm family: SURVEY 8(d)), and the heaviest one on the MFMA side (the 128-512-channel levels) next to it
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        suffix = '_%s_pmc_hbm_per_kernel.json' % which
        rl = {s_: kernel_roofline(s_, d, peak, args.profile_steps, overhead, pmc_suffix=suffix, shape=(batch_clips, seq, size, dtype_name))
              for s_, d in by_kernel.items() if s_ and d['bytes'] > 0 and d['ms'] > 0}
        # `roofline` = the heaviest symbol by TIME, whichever roof its intensity puts it under; `roofline_other_bound` = the
        # heaviest symbol on the other side of the ridge
        if rl:
            dom = max(rl, key=lambda s_: by_kernel[s_]['ms'])
            out['roofline'] = rl[dom]
            other = [s_ for s_ in rl if rl[s_]['bound'] != rl[dom]['bound']]
            if other:
                out['roofline_other_bound'] = rl[max(other, key=lambda s_: by_kernel[s_]['ms'])]
        out['kernels_ms_per_step'] = {s_: round(by_kernel[s_]['ms'] / args.profile_steps, 4) for s_ in by_kernel if s_}
        out['kernel_groups_ms_per_step'] = {t: round(prof[t]['ms'] / args.profile_steps, 4) for t in prof}
    del tr
    torch.cuda.empty_cache()
    eve_amd.reset_standalone_config()
    return out


def eyenet_point(args, device, dtype_name, batch, steps, warmup, k, profile=True):
    """One more operating point of the configs[1] workload on this GPU (world == 1): `dtype_name`, `batch` clips per step.
    -> {ms_per_step, value, step MFMA fraction, roofline of its dominant kernel}."""
    import eve_amd
    from eve_amd import train
    cfg = eve_amd.reset_standalone_config()
    cfg.import_json(os.path.join(HERE, 'configs', 'eye_net.json'))
    torch.manual_seed(1234)
    net = eve_amd.EyeNet()
    net.compute_dtype = TORCH_DTYPE[dtype_name]
    net.to(device)
    tr = train.eyenet_trainer(net, cfg, use_graph=not args.no_graph)
    tr.static_inputs = 'alias'
    data = synthetic_eyenet_batch(batch, args.seq, args.size, device, 77)
    for _ in range(warmup):
        tr.step(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        terms = tr.step(data)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    peak = MFMA_PEAK_TFLOPS[dtype_name]
    out = {'dtype': dtype_name, 'batch_per_gpu': batch, 'seq_len': args.seq, 'steps': steps, 'ms_per_step': ms,
           'value': batch * args.seq / (ms * 1e-3), 'unit': 'frames/s', 'final_loss': float(terms['full_loss'].detach())}
    if args.size == 128:
        out['step_algorithmic_tflops'] = EYENET_TRAIN_GFLOP_PER_FRAME_128 * out['value'] / 1e3
        out['step_mfma_frac'] = out['step_algorithmic_tflops'] / peak
    if profile and not args.no_roofline:
        tr._eager_step(data)
        torch.cuda.synchronize()
        k.start_profile()
        for _ in range(args.profile_steps):
            tr._eager_step(data)
        prof = k.stop_profile()
        overhead = prof.pop('_event_overhead_ms', None)
        by_kernel = prof.pop('_by_kernel', {})
        if by_kernel:
            dom = max(by_kernel, key=lambda t: by_kernel[t]['ms'])
            d = by_kernel[dom]
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': 'eve::' + dom, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                               'frac': ach / peak, 'traffic': None, 'launches_per_step': d['launches'] / args.profile_steps,
                               'avg_launch_ms': d['ms'] / d['launches'],
                               'algorithmic_gflop_per_launch': d['flops'] / d['launches'] / 1e9,
                               'event_pair_overhead_ms_subtracted': overhead}
    del tr, net
    torch.cuda.empty_cache()
    eve_amd.reset_standalone_config()
    return out


LONG_KEYS = ('kernels_ms_per_step', 'kernel_groups_ms_per_step', 'kernel_groups_tflops', 'roofline_other_bound', 'cpu_baseline',
             'optimizer', 'workload')


def order_line(out):
    """The JSON line ordered for a reader AND for a log tail: the contract keys, `roofline` and `cpu_baseline` lead; the operating
    points follow with their long per-kernel maps moved to the end of each object; `summary` -- one short object with every
    operating point's {ms per step, frames/s, fraction of the MFMA peak, its dominant kernel's roofline fraction} -- is the LAST
    key of the line, because the driver's record keeps a parsed subset plus the last ~1 000 characters."""
    points = ('c3', 'c3_clstm', 'c5', 'b8', 'fp32', 'fp16')
    summary = {}
    for key in points:
        if key in out:
            o = out[key]
            summary[key] = {'ms': round(o['ms_per_step'], 3), 'fps': round(o['value'])}
            if 'step_mfma_frac' in o:
                summary[key]['mfma'] = round(o['step_mfma_frac'], 4)
            if 'roofline' in o:
                summary[key]['roof'] = '%s %.3f' % (o['roofline'].get('bound'), o['roofline'].get('frac'))

    def tail_long(o):
        if not isinstance(o, dict):
            return o
        return dict([(k_, v) for k_, v in o.items() if k_ not in LONG_KEYS] + [(k_, o[k_]) for k_ in LONG_KEYS if k_ in o])
    head = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline']
    line = {k_: out[k_] for k_ in head if k_ in out}
    for k_, v in out.items():
        if k_ not in line and k_ not in points and k_ not in LONG_KEYS:
            line[k_] = v
    for k_ in LONG_KEYS:
        if k_ in out and k_ not in line:
            line[k_] = out[k_]
    for k_ in points:
        if k_ in out:
            line[k_] = tail_long(out[k_])
    if summary:
        groups = out.get('kernel_groups_tflops') or {}
        if groups:
            summary['tflops'] = {k_: round(v) for k_, v in groups.items()}
        line.pop('summary', None)
        line['summary'] = summary
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3', 'c5'],
                    help="what the MAIN line measures: c2 = BASELINE configs[1] (EyeNet training, the quoted metric; default), "
                         "c3 = configs[2]/[3] (EyeNet + RefineNet pipeline; with --gpus N this is configs[3]), c5 = configs[4] "
                         "(T=120, 256x256, fp16, both networks trained; defaults --batch 8 --seq 120 --size 256 --dtype fp16)")
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU (default 32; 8 for --workload c5)')
    ap.add_argument('--seq', type=int, default=None)
    ap.add_argument('--size', type=int, default=None)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a hipGraph')
    ap.add_argument('--graph', action='store_true', help='(kept for compatibility: hipGraph replay is the default at every world size)')
    ap.add_argument('--graph-collectives', action='store_true',
                    help='several ranks: capture the bucket all-reduces, the clip and Adam into the hipGraph as well (RCCL)')
    ap.add_argument('--profile-steps', type=int, default=2)
    ap.add_argument('--no-c3', action='store_true', help='skip the configs[2] (EyeNet + RefineNet pipeline) measurement')
    ap.add_argument('--no-c5', action='store_true', help='skip the configs[4] (T=120, 256x256, fp16) measurement')
    ap.add_argument('--no-points', action='store_true', help='skip the extra operating points (fp32 parity mode, B=8 per GPU, fp16)')
    args = ap.parse_args()
    c5 = args.workload == 'c5'
    args.batch = args.batch if args.batch is not None else (8 if c5 else 32)
    args.seq = args.seq if args.seq is not None else (120 if c5 else 30)
    args.size = args.size if args.size is not None else (256 if c5 else 128)
    args.dtype = args.dtype if args.dtype is not None else ('fp16' if c5 else 'bf16')

    if args.graph_collectives:
        os.environ['EVE_AMD_GRAPH_COLLECTIVES'] = '1'
    import eve_amd
    from eve_amd import parallel, train
    from eve_amd.kernels import default_kernels
    import warnings
    # (the capture's warm-up runs on a side stream by design; torch's per-process warning about it would fill the log tail)
    warnings.filterwarnings('ignore', message=".*AccumulateGrad node's stream does not match.*")
    if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    dev_index = int(os.environ.get('EVE_AMD_FORCE_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    k = default_kernels()
    # ONE execution mode at every world size, so that the 1 -> N curve compares like with like: forward + backward replay as a
    # hipGraph; with several ranks the bucket all-reduces (RCCL), the clip and Adam are launched eagerly behind each replay
    # (tests/test_gpu_data_parallel.py runs exactly this with two ranks), or captured too with --graph-collectives.
    use_graph = not args.no_graph
    # (EVE_AMD_FORCE_DIST=1: a one-rank RCCL group on a one-GPU box -- the transport and the bucket launches with world = 1)
    distributed = world > 1 or os.environ.get('EVE_AMD_FORCE_DIST', '0') == '1'

    if args.workload == 'c2':
        cfg = eve_amd.reset_standalone_config()
        cfg.import_json(os.path.join(HERE, 'configs', 'eye_net.json'))
        torch.manual_seed(1234)
        net = eve_amd.EyeNet()
        net.compute_dtype = TORCH_DTYPE[args.dtype]
        net.to(device)
        trainer = train.eyenet_trainer(net, cfg, distributed=distributed, use_graph=use_graph)
        trainer.static_inputs = 'alias'      # the synthetic batch stays in the same device buffers: the graph reads it in place
        batch = synthetic_eyenet_batch(args.batch, args.seq, args.size, device, 1000 * rank)
        gflop_per_frame = EYENET_TRAIN_GFLOP_PER_FRAME_128 if args.size == 128 else None
        workload = ('BASELINE configs[1]: EyeNet training (configs/eye_net.json: ResNet-18-IN + GRU-128, angular + pupil L1 losses, '
                    'clip 5.0, Adam wd 0.005), %dx%d patches, fwd+bwd+clip+Adam' % (args.size, args.size))
    else:
        import numpy as np
        np.random.seed(1000 * rank)          # per-rank kappa_fake streams
        trainer, batch, cfg = pipeline_setup(args, device, args.workload, args.batch, args.seq, args.size, args.dtype, use_graph,
                                             distributed=distributed, seed=1 + rank)
        net = None
        gflop_per_frame = PIPELINE_GFLOP.get((args.workload, args.size))
        workload = ('BASELINE configs[%s]: eve_amd.EVE pipeline (%s), %dx%d patches, fwd+bwd+clip+Adam'
                    % ('2' if args.workload == 'c3' and world == 1 else ('3' if args.workload == 'c3' else '4'),
                       'EyeNet frozen fwd + RefineNet/CGRU trained' if args.workload == 'c3' else 'EyeNet + RefineNet/CGRU trained',
                       args.size, args.size))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        terms = trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = trainer.step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch kernel durations for the roofline: the same step launched eagerly with a HIP event pair around
    # every conv launch (a graph replay has no per-kernel events; the kernels and shapes are identical)
    prof, event_overhead = {}, None
    if not args.no_roofline:
        trainer._eager_step(batch)           # settle the caching allocator outside the captured graph's pool
        torch.cuda.synchronize()
        k.start_profile()
        for _ in range(args.profile_steps):
            trainer._eager_step(batch)
        prof = k.stop_profile()
        event_overhead = prof.pop('_event_overhead_ms', None)
        by_kernel = prof.pop('_by_kernel', {})
        barrier()
    loss = float(terms['full_loss'].detach())
    ranks_seen = 1
    if world > 1:        # proof that the collective saw every rank (RCCL over xGMI): an all-reduce of ones
        ones = torch.ones(1, dtype=torch.float32, device=device)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(round(float(ones)))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = 1e3 * elapsed / args.steps
    frames = world * args.batch * args.seq
    value = frames / (elapsed / args.steps)

    if rank == 0:
        out = {
            'metric': 'train-step frames/sec, %dx%d eye patches T=%d' % (args.size, args.size, args.seq),
            'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload, 'global_batch': world * args.batch, 'batch_per_gpu': args.batch, 'seq_len': args.seq,
                       'parallelism': 'dp%d' % world},
            'final_loss': loss, 'hip_graph': use_graph, 'graph_collectives': bool(getattr(trainer, 'graph_collectives', False)),
            'collectives': None if not distributed else ('captured in the hipGraph' if getattr(trainer, 'graph_collectives', False) else
                                                    ('eager RCCL launches behind gate kernels released by signal nodes of the running hipGraph replay (overlapped with backward)' if use_graph else 'eager, overlapped with backward')),
            'ranks_seen': ranks_seen, 'optimizer': trainer.optimizer_state(),
            'gate_timeouts': (trainer.sync.gate_timeouts() if getattr(trainer, 'sync', None) is not None else None),
            'comm_stream_runs_beside_replay': (trainer.sync.overlaps if getattr(trainer, 'sync', None) is not None else None),
            'kernel_tree_sha': __import__('eve_amd.build', fromlist=['kernel_tree_sha']).kernel_tree_sha(),
            'dispatch_config_is_default': k.dispatch_config().as_dict() == k.default_dispatch_config().as_dict(),
        }
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        if gflop_per_frame is not None:
            out['step_algorithmic_tflops'] = gflop_per_frame * value / 1e3 / world
            out['step_mfma_frac'] = out['step_algorithmic_tflops'] / peak
        if prof:
            # the dominant KERNEL (one symbol = one row of the rocprofv3 kernel summary in profiles/), by total time
            dom = max(by_kernel, key=lambda t: by_kernel[t]['ms'])
            sfx = None if args.workload == 'c2' else '_%s_pmc_hbm_per_kernel.json' % args.workload
            shape = (args.batch, args.seq, args.size, args.dtype)
            out['roofline'] = kernel_roofline(dom, by_kernel[dom], peak, args.profile_steps, event_overhead, pmc_suffix=sfx, shape=shape)
            # ... and the heaviest kernel on the OTHER side of the ridge, so that both roofs are on the line
            for t in sorted(by_kernel, key=lambda t: -by_kernel[t]['ms']):
                r = kernel_roofline(t, by_kernel[t], peak, args.profile_steps, event_overhead, pmc_suffix=sfx, shape=shape)
                if r['bound'] != out['roofline']['bound'] and by_kernel[t]['flops'] > 0:
                    out['roofline_other_bound'] = r
                    break
            out['kernels_ms_per_step'] = {t: round(by_kernel[t]['ms'] / args.profile_steps, 4) for t in by_kernel}
            out['kernel_groups_ms_per_step'] = {t: prof[t]['ms'] / args.profile_steps for t in prof}
            out['kernel_groups_tflops'] = {t: prof[t]['flops'] / (prof[t]['ms'] * 1e-3) / 1e12 for t in prof if prof[t]['flops'] > 0}
        main_c2 = args.workload == 'c2' and args.size == 128
        extras = world == 1 and main_c2 and args.batch == 32 and args.dtype == 'bf16'
        if world == 1 and main_c2 and not args.no_c3:
            del trainer, net                      # release the EyeNet trainer's graph pool before the second workload
            torch.cuda.empty_cache()
            out['c3'] = bench_pipeline(args, device, k, 'c3', args.batch, args.seq, 128, args.dtype, max(3, args.steps // 2), args.warmup)
            if extras and not args.no_points:
                # the reference's SHIPPED cell (src/configs/refine_net.json:55 "CLSTM"; its hidden state is computed and stored,
                # the features pass through: refine_net.py:168-174), same pipeline otherwise
                out['c3_clstm'] = bench_pipeline(args, device, k, 'c3', args.batch, args.seq, 128, args.dtype, 3, 2, rnn='CLSTM', profile=False)
        if extras and not args.no_c5:
            try:
                del trainer, net
            except NameError:
                pass
            torch.cuda.empty_cache()
            # BASELINE configs[4] on this GPU: B = 8 clips x T = 120 = the same 1 920 patches / 960 frames per step as configs[1]
            out['c5'] = bench_pipeline(args, device, k, 'c5', 8, 120, 256, 'fp16', 3, 2)
        if extras and not args.no_points:
            # the other stated operating points, same workload: north_star's B = 8 clips per GPU, and the float32 parity
            # mode (the instantiation that holds the 1e-4 rad tolerance; its roofline is the 157.3 TFLOP/s f32-input MFMA peak)
            try:
                del trainer, net
            except NameError:
                pass
            torch.cuda.empty_cache()
            out['b8'] = eyenet_point(args, device, 'bf16', 8, max(args.steps, 10), args.warmup, k)
            out['fp32'] = eyenet_point(args, device, 'fp32', args.batch, max(3, args.steps // 2), 2, k)
            out['fp16'] = eyenet_point(args, device, 'fp16', args.batch, args.steps, args.warmup, k, profile=False)
        if world == 1 and not args.no_cpu_baseline:
            # (`cores` = the threads actually used: CPU_BASELINE_THREADS; `host_cpus` = what the box has)
            out['cpu_baseline'] = cpu_baseline(args.seq if args.workload == 'c2' else 30, args.size if args.workload == 'c2' else 128)
            if 'c3' in out:
                out['c3']['cpu_baseline'] = cpu_baseline_c3(args.seq)
        print(json.dumps(order_line(out)), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
