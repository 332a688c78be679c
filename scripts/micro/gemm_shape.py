"""one pointwise-GEMM shape in its three roles, back-to-back launches (rotating buffers): python gemm_shape.py M K N
(used with the grid knobs DL3P_GEMM_MI / DL3P_GEMM_NT_MAX / DL3P_GEMM_PER_CU / DL3P_WGRAD_PER_CU)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
if os.environ.get('DL3P_LIB_VARIANT'):          # A/B against a library built by build_variant.sh
    libm = importlib.import_module('tf-keras-deeplabv3p-model-set_amd._lib')
    libm._lib = libm.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdl3p_%s.so' % os.environ['DL3P_LIB_VARIANT']))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
M, K, N = (int(a) for a in sys.argv[1:4])
NB = 4
xs = [torch.randn(M, K, device='cuda') for _ in range(NB)]
gs = [torch.randn(M, N, device='cuda') for _ in range(NB)]
w = torch.randn(K, N, device='cuda') / K ** 0.5
wt = w.t().contiguous()
sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda')
part = ops.new_partials(max(K, N), 'cuda')
ys = [torch.empty(M, N, device='cuda') for _ in range(NB)]
gxs = [torch.empty(M, K, device='cuda') for _ in range(NB)]
i = [0]


def timeit(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def fwd():
    i[0] = (i[0] + 1) % NB
    ops.pwconv_fwd_wt(xs[i[0]], wt, None, sc, sh, ops.ACT_RELU, out=ys[i[0]], partials=part)


def dgrad():
    i[0] = (i[0] + 1) % NB
    ops.pwconv_bwd_data(gs[i[0]], w, out=gxs[i[0]])


def wgrad():
    i[0] = (i[0] + 1) % NB
    ops.pwconv_bwd_weight(xs[i[0]], gs[i[0]], sc, sh, ops.ACT_RELU)


zs = [torch.randn(M, K, device='cuda') for _ in range(2)]
mean, invstd = torch.zeros(K, device='cuda'), torch.ones(K, device='cuda')


def dgrad_bn():
    # the data gradient with the fused BatchNorm-backward sums (what the executor launches behind a BN + activation)
    i[0] = (i[0] + 1) % NB
    ops.pwconv_bwd_data_bn(gs[i[0]], w, zs[i[0] % 2], sc, sh, ops.ACT_RELU6, mean, invstd, part, out=gxs[i[0]])


fl = 2.0 * M * K * N / 157e6
print('M=%d K=%d N=%d  floor %.1f us (157 TF)   fwd %.1f us  dgrad %.1f us  dgrad+bn %.1f us  wgrad %.1f us' % (M, K, N, fl, timeit(fwd), timeit(dgrad), timeit(dgrad_bn), timeit(wgrad)))
