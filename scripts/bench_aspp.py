import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
N, H, W, C = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 33, 33, int(sys.argv[2]) if len(sys.argv) > 2 else 320
def timeit(f, R=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
x = torch.randn((N, H, W, C), device='cuda'); y = torch.empty_like(x)
w = torch.randn((3, 3, C), device='cuda'); sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda')
part = ops.new_partials(C, 'cuda')
byt = 2 * x.numel() * 4
for rate in (18, 12, 6, 2, 1):
    a = timeit(lambda: ops.dwconv2d_fwd(x, w, 1, rate, 'same', sc, sh, ops.ACT_NONE, out=y, partials=part))
    b = timeit(lambda: ops.dwconv2d_fwd(x, w, 1, rate, 'same', sc, sh, ops.ACT_NONE, out=y))
    c = timeit(lambda: ops.dwconv2d_fwd(x, w, 1, rate, 'same', out=y))
    print('rate %2d: full %.1f us (%.0f GB/s) | no stats %.1f | no stats no prologue %.1f' % (rate, a, byt / a / 1e3, b, c))
print('copy %.1f us' % timeit(lambda: y.copy_(x)), ' affine_act %.1f us' % timeit(lambda: ops.affine_act(x, sc, sh, ops.ACT_RELU, out=y)))
if hasattr(ops, 'aspp_dw3_fwd'):
    ys = [torch.empty_like(x) for _ in range(3)]
    ws = torch.randn((3, 3, 3, C), device='cuda')
    parts = torch.empty(3 * part.numel(), device='cuda')
    a = timeit(lambda: ops.aspp_dw3_fwd(x, ws, (6, 12, 18), sc, sh, ops.ACT_NONE, ys, parts))
    print('fused 3-rate: %.1f us (%.0f GB/s algorithmic 4 tensors)' % (a, 2 * byt / a / 1e3))
