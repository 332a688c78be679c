import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
def timeit(f, R=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
for M, C in [(16*129*129, 304), (16*129*129, 256), (16*33*33, 320), (16*257*257, 32), (16*257*257, 96)]:
    x = torch.randn((M, C), device='cuda'); y = torch.empty_like(x)
    sc = torch.rand(C, device='cuda'); sh = torch.rand(C, device='cuda')
    us = timeit(lambda: ops.affine_act(x, sc, sh, ops.ACT_RELU6, out=y))
    us2 = timeit(lambda: y.copy_(x))
    us3 = timeit(lambda: torch.clamp(x * 1.5 + 0.5, 0, 6, out=y) if False else torch.add(x, 1.0, out=y))
    print('M=%d C=%d affine_act %.1f us %.0f GB/s | torch copy %.1f us %.0f GB/s | torch add %.1f us %.0f GB/s' % (M, C, us, 2*M*C*4/us/1e3, us2, 2*M*C*4/us2/1e3, us3, 2*M*C*4/us3/1e3))
