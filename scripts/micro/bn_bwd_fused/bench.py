"""dl3p_bn_bwd_fused against the reduce + finalize + apply chain it replaces, per shape (us per BatchNorm backward, hipGraph replay of 20
back-to-back instances so that launch latency is what a graph step pays)."""
import importlib
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tests.conftest import load_pkg       # noqa: E402
ops = importlib.import_module(load_pkg().__name__ + '.ops')
DEV = 'cuda:0'
CASES = [(64 * 128, 960, True), (64 * 128, 672, True), (64 * 128, 160, True), (128 * 256, 240, True), (128 * 256, 40, True), (256 * 512, 64, True),
         (256 * 512, 24, True), (512 * 1024, 16, True), (16 * 33 * 33, 256, False), (16 * 33 * 33, 320, False), (4 * 33 * 33, 728, False), (4 * 33 * 33, 256, False)]


def timed(fn, reps=20, iters=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * iters)


for M, C, bf16 in CASES:
    dt = torch.bfloat16 if bf16 else torch.float32
    z = (torch.randn(M, C, device=DEV) * 2).to(dt)
    g = torch.randn(M, C, device=DEV).to(dt)
    out = torch.empty_like(g)
    bn = ops.BNState(C, DEV)
    bn.mean.zero_(); bn.invstd.fill_(0.5); bn.scale.fill_(0.5); bn.shift.zero_()
    part = ops.new_partials(C, DEV)
    if not ops.bn_backward_fused_supported(z, bf16):
        print('%7d x %4d %s: not served' % (M, C, dt)); continue
    ops.bn_fused_workspace(torch.device(DEV))

    def chain():
        gg = out
        gg.copy_(g) if False else None
        (ops.bn_backward_bf16(bn, g, z, 2, part) if bf16 else ops.bn_backward(bn, g, z, 2, part, out=out))
    t_chain = timed(chain)
    t_fused = timed(lambda: ops.bn_backward_fused(bn, g, z, 2, out=out))
    mb = M * C * (2 if bf16 else 4) / 1e6
    print('%7d x %4d %-8s %6.1f MB   chain %6.1f us   fused %6.1f us' % (M, C, 'bf16' if bf16 else 'f32', mb, t_chain, t_fused), flush=True)
