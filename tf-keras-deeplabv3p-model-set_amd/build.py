"""Build libdl3p.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  One object per .hip file (parallel), then one shared library next
to this file so that it travels with the source tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libdl3p.so')
OBJDIR = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off']
# the fused inverted-residual kernels are bound by vector-instruction issue: MFMA results straight into VGPRs (no v_accvgpr_read per
# result register)
EXTRA = {'irb_fwd.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'irb_bwd.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'dl3p.h'))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + EXTRA.get(src, []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stderr))
        return r
    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
