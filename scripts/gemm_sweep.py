"""kernel-precise timing (dl3p_probe_arm) of the pointwise GEMM entry points on the model's layer shapes
   python scripts/gemm_sweep.py [fwd|dgrad|wgrad|all] [shape-subset: all|small|big]"""
import sys, os, importlib, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
L = ops.lib()
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
subset = sys.argv[2] if len(sys.argv) > 2 else 'all'
big = [(1056784, 32, 32), (1056784, 32, 16), (1056784, 16, 96), (266256, 96, 24), (266256, 24, 144), (266256, 144, 24),
       (266256, 304, 256), (266256, 256, 256), (266256, 256, 24)]
small = [(67600, 144, 32), (67600, 32, 192), (67600, 192, 32), (17424, 192, 64), (17424, 64, 384), (17424, 384, 64),
         (17424, 384, 96), (17424, 96, 576), (17424, 576, 96), (17424, 576, 160), (17424, 160, 960), (17424, 960, 160),
         (17424, 960, 320), (17424, 320, 256), (17424, 1280, 256)]
shapes = {'all': big + small, 'big': big, 'small': small}[subset]
HBM, MFMA = 6.3e6, 157e6
slot = [0]
def probe(f, R=7):
    for _ in range(2): f()
    ts = []
    for _ in range(R):
        L.probe_arm(slot[0]); f()
        ms = ctypes.c_float(0); L.probe_read(slot[0], ctypes.addressof(ms)); ts.append(ms.value * 1e3)
        slot[0] = (slot[0] + 1) % 4096
    ts.sort()
    return ts[len(ts) // 2]
tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0, 'floor': 0.0}
for M, K, N in shapes:
    x = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda') * 0.05
    sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda') * 0.1
    y = torch.empty((M, N), device='cuda'); dy = torch.randn((M, N), device='cuda'); gx = torch.empty((M, K), device='cuda')
    part = ops.new_partials(N, 'cuda')
    ws = torch.empty(L.pwconv_bwd_weight_workspace(M, K, N) // 4, device='cuda')
    fl = max(4.0 * M * (K + N) / HBM, 2.0 * M * K * N / MFMA)
    line = 'M=%7d K=%4d N=%4d floor %6.1f us |' % (M, K, N, fl)
    tot['floor'] += fl
    if which in ('all', 'fwd'):
        t = probe(lambda: ops.pwconv_fwd(x, w, None, sc, sh, ops.ACT_RELU6, out=y, partials=part)); tot['fwd'] += t
        line += ' fwd %6.1f (%3.0f%%)' % (t, 100 * fl / t)
    if which in ('all', 'dgrad'):
        t = probe(lambda: ops.pwconv_bwd_data(dy, w, out=gx)); tot['dgrad'] += t
        line += ' dgrad %6.1f (%3.0f%%)' % (t, 100 * fl / t)
    if which == 'wgrad_ev':      # whole entry point (kernel + slab reduce) by events over back-to-back calls
        f = lambda: ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU6, workspace=ws)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 20; tot['wgrad'] += t
        line += ' wgrad(total) %6.1f (%3.0f%%)' % (t, 100 * fl / t)
    if which in ('all', 'wgrad'):
        t = probe(lambda: ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU6, workspace=ws)); tot['wgrad'] += t
        line += ' wgrad %6.1f (%3.0f%%)' % (t, 100 * fl / t)
    print(line, flush=True)
    del x, y, dy, gx, ws
print('totals', {k: round(v, 1) for k, v in tot.items()})
