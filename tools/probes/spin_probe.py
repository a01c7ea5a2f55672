"""profiles/r06_notes.md section 9: time kernels of the step alone and beside a resident one-wave spinner on another stream.
Build first: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/spin_probe.so tools/probes/spin_probe.hip"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from eve_amd.kernels import HipKernels
k = HipKernels()
sp = ctypes.CDLL(os.path.join(here, 'spin_probe.so'))
P = ctypes.c_void_p
sp.spin_launch.argtypes = [P, ctypes.c_int, ctypes.c_uint, P, ctypes.c_int, P]
sp.spin_set.argtypes = [P, ctypes.c_uint, P]
dev = torch.device('cuda:0')
flag = torch.zeros(4, dtype=torch.int32, device=dev)
out = torch.zeros(4, dtype=torch.int32, device=dev)
N = 1920
src = torch.randn((N, 3, 128, 128), device=dev) + 0.2
w8 = (torch.randn((64, 7, 7, 8), device=dev) * 0.05).to(torch.bfloat16); w8[..., 3:] = 0
xp = k.stem_pack_input(src, dtype=torch.bfloat16)
x64 = torch.randn((N, 64, 32, 32), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if False else None


def stem():
    k.stem_fwd_fused(xp, w8)


def timeit(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.current_stream().synchronize()        # (never the device: the spinner on the side stream must stay resident)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


# a concurrent side stream (probe like parallel.GradSync._pick_concurrent_stream: try until the spinner really co-runs)
MAXIT = 300000


def run(kind, threads, fn, label):
    best = None
    for attempt in range(12):
        side = torch.cuda.Stream(device=dev)
        flag.zero_(); out.zero_()
        torch.cuda.synchronize()
        sp.spin_launch(P(flag.data_ptr()), kind, MAXIT, P(out.data_ptr()), threads, P(side.cuda_stream))
        ms = timeit(fn)
        sp.spin_set(P(flag.data_ptr()), 1, P(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        iters = int(out[0])
        if 0 < iters < MAXIT:     # the spinner was still resident when fn had run (an aliased stream runs it to its bound first)
            best = (ms, iters)
            break
    print('%-28s kind %d threads %3d : %s' % (label, kind, threads, 'spinner never co-ran' if best is None else '%.4f ms (spinner iterations %d)' % best), flush=True)


print('stem_fwd alone: %.4f ms' % timeit(stem), flush=True)
for kind in (0, 1, 2, 3):
    run(kind, 64, stem, 'stem_fwd beside spinner')
run(0, 1, stem, 'stem_fwd beside spinner')
print('stem_fwd alone again: %.4f ms' % timeit(stem), flush=True)
