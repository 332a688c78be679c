"""one bf16 pointwise GEMM shape, repeated (for kernel traces / PMC passes): python scripts/micro/bf16_gemm.py M K N [fwd|dgrad|wgrad] [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
M, K, N = (int(a) for a in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
w = torch.randn(K, N, device='cuda') / K ** 0.5
sc = torch.rand(K, device='cuda') + 0.5
sh = torch.randn(K, device='cuda') * 0.3
gy = torch.randn(M, N, device='cuda').to(torch.bfloat16)
part = ops.new_partials(N, 'cuda')
for _ in range(reps):
    if mode == 'fwd':
        ops.pwconv_fwd_bf16(x, w, None, sc, sh, ops.ACT_RELU, partials=part)
    elif mode == 'fwd_nostats':
        ops.pwconv_fwd_bf16(x, w, None, sc, sh, ops.ACT_RELU)
    elif mode == 'dgrad':
        ops.pwconv_bwd_data_bf16(gy, w)
    else:
        ops.pwconv_bwd_weight_bf16(x, gy, sc, sh, ops.ACT_RELU)
torch.cuda.synchronize()
