import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
N, H, W, C = 16, 33, 33, 320
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
xs = [torch.randn((N, H, W, C), device='cuda') for _ in range(nbuf)]
ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
w = torch.randn((3, 3, C), device='cuda'); sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda')
part = ops.new_partials(C, 'cuda')
for i in range(60):
    ops.dwconv2d_fwd(xs[i % nbuf], w, 1, 18, 'same', sc, sh, ops.ACT_NONE, out=ys[i % nbuf], partials=part)
torch.cuda.synchronize()
