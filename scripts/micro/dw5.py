"""Times the 5x5 depthwise layers of MobileNetV3-large (batch 16, 513x513, OS16) in their three roles, cold-ish
(a 512 MB sweep between launches would be fairer; here the tensors of 5 layers rotate).  usage: python dw5.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
MAXR = 4096


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [(16, 129, 129, 72, 5, 2, 1), (16, 65, 65, 120, 5, 1, 1), (16, 33, 33, 672, 5, 1, 1), (16, 33, 33, 960, 5, 1, 2),
         (16, 33, 33, 960, 3, 1, 1)]
for N, H, W, C, k, s, r in cases:
    NB = 6                                      # rotate buffers so that L2/MALL do not hold everything
    xs = [torch.randn(N, H, W, C, device='cuda') for _ in range(NB)]
    w = torch.randn(k, k, C, device='cuda')
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda')
    y0 = ops.dwconv2d_fwd(xs[0], w, s, r, in_scale=sc, in_shift=sh, in_act=ops.ACT_HSWISH)
    ys = [torch.empty_like(y0) for _ in range(NB)]
    part = torch.zeros(MAXR * 2 * C, device='cuda')
    i = [0]

    def fwd():
        i[0] = (i[0] + 1) % NB
        ops.dwconv2d_fwd(xs[i[0]], w, s, r, in_scale=sc, in_shift=sh, in_act=ops.ACT_HSWISH, out=ys[i[0]], partials=part)

    def bwd_d():
        i[0] = (i[0] + 1) % NB
        ops.dwconv2d_bwd_data(ys[i[0]], w, xs[0].shape, s, r, out=xs[i[0]])

    mean = torch.randn(C, device='cuda'); inv = torch.rand(C, device='cuda') + 0.5
    zs = [torch.randn(N, H, W, C, device='cuda') for _ in range(NB)]
    gx = [torch.empty(N, H, W, C, device='cuda') for _ in range(NB)]

    def bwd_dbn():
        i[0] = (i[0] + 1) % NB
        ops.dwconv2d_bwd_data_bn(ys[i[0]], w, xs[0].shape, zs[i[0]], sc, sh, ops.ACT_HSWISH, mean, inv, part, s, r, out=gx[i[0]])

    def bwd_w():
        i[0] = (i[0] + 1) % NB
        ops.dwconv2d_bwd_weight(xs[i[0]], ys[i[0]], k, s, r, in_scale=sc, in_shift=sh, in_act=ops.ACT_HSWISH)

    mb = (xs[0].numel() + ys[0].numel()) * 4 / 1e6
    print('%-28s %6.1f MB  fwd %7.1f us  bwd_data %7.1f us  bwd_data_bn %7.1f us  bwd_weight %7.1f us   (floor %.1f us)' % (
        str((N, H, W, C, k, s, r)), mb, timeit(fwd), timeit(bwd_d), timeit(bwd_dbn), timeit(bwd_w), mb / 6.3))
