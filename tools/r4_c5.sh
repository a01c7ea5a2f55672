#!/bin/bash
# c5 (T=120, 256x256, fp16): big-plane InstanceNorm on / off, instnorm tests, IN microbench at the 256x256 trunk planes
mkdir -p gpurun_out/c5w
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "instnorm" > gpurun_out/c5w/in_tests.log 2>&1
tail -3 gpurun_out/c5w/in_tests.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, torch
sys.path.insert(0, 'tools')
from bench_in import timeit
from eve_amd.kernels import HipKernels
k = HipKernels()
dt = torch.float16
for name, N, H, C in (('L1 256', 960, 64, 64), ('L2 256', 960, 32, 128)):
    x = torch.randn((N, H, H, C), device='cuda').to(dt); r = torch.randn_like(x); dy = torch.randn_like(x); dy2 = torch.randn_like(x)
    mb = x.numel() * 2 / 1e6
    for big in (0, 1):
        with k.dispatch_override(in_big_planes=big):
            def fwd(res=None, mask=False):
                out = k.instnorm_fwd_fused(x, None, None, res, 1, want_mask=mask)
                if out is None:
                    mr = k.instnorm_stats(x, 1e-5)
                    return (k.instnorm_act_fwd(x, mr, None, None, res, 1), mr, None)
                return out
            t1 = timeit(lambda: fwd()); t2 = timeit(lambda: fwd(r, True))
            y, mr, mask = fwd(r, True)
            def bwd_mid():
                out = k.instnorm_bwd_fused(dy, None, x, mr, None, 1, False)
                return out if out is not None else k.instnorm_act_bwd(dy, None, x, mr, None, 1, False)
            def bwd_end():
                out = k.instnorm_bwd_fused(dy, None if mask is not None else y, x, mr, None, 1, True, mask=mask, dy2=dy2) if mask is not None else None
                if out is None:
                    out = k.instnorm_act_bwd(k.add(dy, dy2), y, x, mr, None, 1, True)
                return out
            t3 = timeit(bwd_mid); t4 = timeit(bwd_end)
            print('%s big=%d fwd %.3f (%.2f TB/s) fwd+res %.3f (%.2f)  bwd mid %.3f (%.2f)  bwd end %.3f (%.2f)  %s' % (
                name, big, t1, 2*mb/t1/1e3, t2, 3*mb/t2/1e3, t3, 3*mb/t3/1e3, t4, 5*mb/t4/1e3, k.lib.eve_last_kernel().decode()[:50]))
PY
for big in 1 0; do
EVE_IN_BIG_PLANES=$big python bench.py --workload c5 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 big=$big', d['value'], d['ms_per_step'])"
done
