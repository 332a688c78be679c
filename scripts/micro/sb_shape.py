"""one pointwise-GEMM shape on the split-bf16 kernels (forward + statistics, data gradient), back-to-back launches on rotating buffers,
for counter passes (GEMM_PMC_SCRIPT=scripts/micro/sb_shape.py bash scripts/micro/gemm_pmc.sh "M K N"): python sb_shape.py M K N"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
L = importlib.import_module('tf-keras-deeplabv3p-model-set_amd._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
M, K, N = (int(a) for a in sys.argv[1:4])
NB = 4
xs = [torch.randn(M, K, device='cuda') for _ in range(NB)]
gs = [torch.randn(M, N, device='cuda') for _ in range(NB)]
w = torch.randn(K, N, device='cuda') / K ** 0.5
wsp_f, wsp_b = ops.split_bf16x3(w.t().contiguous()), ops.split_bf16x3(w.contiguous())
sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda')
part = ops.new_partials(max(K, N), 'cuda')
ys = [torch.empty(M, N, device='cuda') for _ in range(NB)]
gxs = [torch.empty(M, K, device='cuda') for _ in range(NB)]
for r in range(40):
    i = r % NB
    ops.pwconv_fwd_sb(xs[i], wsp_f, K, None, sc, sh, ops.ACT_RELU, out=ys[i], partials=part)
for r in range(40):
    i = r % NB
    ops.pwconv_bwd_data_sb(gs[i], wsp_b, N, out=gxs[i])
torch.cuda.synchronize()
