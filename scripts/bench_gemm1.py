import sys, os, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
M, K, N = 266256, 256, 256
x = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda') * 0.05
dy = torch.randn((M, N), device='cuda'); gx = torch.empty((M, K), device='cuda')
for _ in range(12):
    ops.pwconv_bwd_data(dy, w, out=gx)
torch.cuda.synchronize()
