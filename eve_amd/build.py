"""Build libeve_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Every csrc/*.hip is compiled to its own object (in parallel, re-used while neither the source nor any header
changed) and the objects are linked into eve_amd/lib/libeve_hip.so."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')
LIB_PATH = os.path.join(LIB_DIR, 'libeve_hip.so')
# -Wno-inline-asm: every LDS-DMA statement names m0 as a clobber (it writes m0 itself, inside the same statement as the
# buffer_load ... lds that reads it) and clang warns "clobber list contains reserved registers: m0" once per statement (~1 700
# times per build).  The contract behind that -- and the compiler-version guard that pins it -- is in csrc/lds_dma.h.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-inline-asm'] + os.environ.get('EVE_HIPCC_FLAGS', '').split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(HERE), 'include', 'eve_hip.h')]


def kernel_tree_sha():
    """sha256 over the kernel sources (csrc/*, include/eve_hip.h): what a profile was taken on.  bench.py only quotes PMC
    traffic from a profiles/ summary that carries the same value (.git does not travel to the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(sources() + headers()):
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def build(force=False, verbose=True, variant=None, extra_flags=()):
    """hipcc --offload-arch=gfx950 -O3 -c csrc/X.hip (each) ; hipcc -shared -> eve_amd/lib/libeve_hip.so
    variant: a tuning experiment's second library, eve_amd/lib/libeve_hip_<variant>.so built with `extra_flags` on top (objects
    in lib/obj_<variant>/); load it for an A/B on one box with EVE_HIP_LIB=<path> (eve_amd/_lib.py).  Not the product build."""
    if variant:
        return _build_into(os.path.join(LIB_DIR, 'libeve_hip_%s.so' % variant), os.path.join(LIB_DIR, 'obj_' + variant), True, verbose,
                           list(extra_flags))
    if not force and not needs_build():
        return LIB_PATH
    return _build_into(LIB_PATH, OBJ_DIR, force, verbose, [])


def _build_into(LIB_PATH, OBJ_DIR, force, verbose, extra):
    FLAGS = globals()['FLAGS'] + extra
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdr_time = max(os.path.getmtime(p) for p in headers())
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        stale = force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)
        if stale:
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, p.returncode, p.stdout

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        for cmd, rc, out in ex.map(run, jobs):
            if rc != 0:
                raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), out))
    stale_objs = set(glob.glob(os.path.join(OBJ_DIR, '*.o'))) - set(objs)
    for o in stale_objs:                   # a source file was removed
        os.remove(o)
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    return LIB_PATH


if __name__ == '__main__':
    import sys
    if '--variant' in sys.argv:          # python -m eve_amd.build --variant NAME -flag ...
        i = sys.argv.index('--variant')
        print(build(variant=sys.argv[i + 1], extra_flags=sys.argv[i + 2:]))
    else:
        build(force='--force' in sys.argv)
