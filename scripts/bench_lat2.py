"""The roofline kernel (rate-18 atrous depthwise, 33x33x320, dl3p_dwconv2d_fwd) outside the training step:
  * at the bench shape (N=16: 22.3 MB in + 22.3 MB out, fits the 256 MB Infinity Cache) with 1 and 8 rotating buffers
  * SURVEY.md section 8d's streaming variant (N=256: 357 MB each way), which cannot be cache-resident
usage: python scripts/bench_lat2.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
H, W, C = 33, 33, 320
w = torch.randn((3, 3, C), device='cuda'); sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda')
part = ops.new_partials(C, 'cuda')


def run(N, nbuf, reps=40):
    xs = [torch.randn((N, H, W, C), device='cuda') for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    for i in range(2 * nbuf):
        ops.dwconv2d_fwd(xs[i % nbuf], w, 1, 18, 'same', sc, sh, ops.ACT_NONE, out=ys[i % nbuf], partials=part)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.dwconv2d_fwd(xs[i % nbuf], w, 1, 18, 'same', sc, sh, ops.ACT_NONE, out=ys[i % nbuf], partials=part)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    by = 2 * N * H * W * C * 4 + 9 * C * 4
    print('N=%3d  %d buffer(s)  %7.1f MB algorithmic  %7.1f us per launch (back-to-back, launch gaps included)  '
          '%.2f TB/s = %.2f of 8 TB/s' % (N, nbuf, by / 1e6, us, by / us / 1e6, by / us / 1e6 / 8))


run(16, 1)
run(16, 8)
run(256, 1, reps=10)
run(256, 3, reps=12)
