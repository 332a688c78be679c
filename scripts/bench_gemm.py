import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
shapes = [(266256, 304, 256), (266256, 256, 256), (17424, 960, 160), (17424, 160, 960), (17424, 320, 256), (17424, 1280, 256),
          (1056784, 16, 96), (1056784, 32, 16), (266256, 144, 24), (67600, 192, 32)]
def timeit(f, R=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / R
for M, K, N in shapes:
    x = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda') * 0.05
    sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda') * 0.1
    y = torch.empty((M, N), device='cuda'); dy = torch.randn((M, N), device='cuda'); gx = torch.empty((M, K), device='cuda')
    part = ops.new_partials(N, 'cuda')
    ws = torch.empty(ops.lib().pwconv_bwd_weight_workspace(M, K, N) // 4, device='cuda')
    tf = timeit(lambda: ops.pwconv_fwd(x, w, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))
    td = timeit(lambda: ops.pwconv_bwd_data(dy, w, out=gx))
    tw = timeit(lambda: ops.pwconv_bwd_weight(x, dy, sc, sh, ops.ACT_RELU6, workspace=ws))
    gf = 2.0 * M * K * N / 1e6
    print('M=%7d K=%4d N=%4d  fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF' % (M, K, N, tf, gf / tf, td, gf / td, tw, gf / tw))
