"""the RGB stem at the headline shape (16 x 513 x 513 x 3 -> 257 x 257 x 32): forward with statistic rows, weight gradient, weight
gradient with the folded BatchNorm apply -- for scripts/pmc_kernel.sh / ktrace.sh"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
dev = 'cuda'
N = int(os.environ.get('STEM_N', 16))
x = torch.rand(N, 513, 513, 3, device=dev) * 2 - 1
w = torch.randn(3, 3, 3, 32, device=dev) * 0.2
part = ops.new_partials(32, dev)
for _ in range(12):
    z, rows = ops.stem_conv_fwd(x, w, 'same', partials=part)
bn = ops.BNState(32, dev)
ops.bn_finalize(bn, part, rows, z.numel() // 32)
g = torch.randn_like(z)
dz = ops.bn_backward(bn, g, z, ops.ACT_RELU6, part, out=torch.empty_like(g))
for _ in range(12):
    ops.stem_conv_bwd_weight(x, dz, 'same')
    ops.stem_conv_bwd_weight_bn(x, g, z, bn, ops.ACT_RELU6, 'same')
torch.cuda.synchronize()
