"""fused inverted-residual kernels (csrc/irb_fwd.hip, irb_bwd.hip) at the BASELINE configs[1] launch shapes, next to the unfused
kernels they replace; knobs: IRB_CT (channel tiles per wave of the forward), IRB_WAVES, IRB_A_WAVES, IRB_B_WGS"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
L = ops.lib()
N = int(os.environ.get('IRB_N', '16'))
SHAPES = [(N, 257, 257, 16, 96, 2), (N, 129, 129, 24, 144, 1), (N, 129, 129, 24, 144, 2), (N, 65, 65, 32, 192, 1),
          (N, 65, 65, 32, 192, 2)]
REPS = 20


def timed(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, H, W, K, C, s) in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] != '%d' % H and sys.argv[1] != 'all':
        continue
    x = torch.randn((n, H, W, K), device='cuda')
    xs = torch.rand(K, device='cuda') + 0.5
    xh = torch.randn(K, device='cuda') * 0.3
    w1 = torch.randn((K, C), device='cuda') / K ** 0.5
    w1t = w1.t().contiguous()
    wdw = torch.randn((3, 3, C), device='cuda') * 0.4
    bn = ops.BNState(C, 'cuda')
    Ho, Wo = -(-H // s), -(-W // s)
    dy = torch.randn((n, Ho, Wo, C), device='cuda')
    part = ops.new_partials(C, 'cuda')
    y = torch.empty((n, Ho, Wo, C), device='cuda')
    gx = torch.empty((n, H, W, K), device='cuda')

    def stats():
        sums = ops.irb_cov_sums(x, xs, xh, ops.ACT_NONE)
        ops.irb_bn_finalize_cov(bn, sums, w1, n * H * W)
    t_stats = timed(stats)
    res = ['%dx%dx%dx%d->%d s%d' % (n, H, W, K, C, s), 'cov+finalize %.1f' % t_stats]
    for ct in ([int(os.environ['IRB_CT'])] if 'IRB_CT' in os.environ else [1, 2, 3]):
        for waves in ([int(os.environ['IRB_WAVES'])] if 'IRB_WAVES' in os.environ else [2048, 4096, 8192]):
            L.irb_set_plan(ct, waves)
            t = timed(lambda: ops.irb_fwd(x, w1, bn.scale, bn.shift, ops.ACT_RELU6, wdw, s, in_scale=xs, in_shift=xh, out=y, partials=part))
            res.append('fwd ct%d w%d %.1f' % (ct, waves, t))
    L.irb_set_plan(0, 0)
    # the unfused pair
    z1 = torch.empty((n, H, W, C), device='cuda')
    p2 = ops.new_partials(C, 'cuda')
    tu1 = timed(lambda: ops.pwconv_fwd_wt(x.view(-1, K), w1t, in_scale=xs, in_shift=xh, out=z1.view(-1, C), partials=p2))
    tu2 = timed(lambda: ops.dwconv2d_fwd(z1, wdw, s, 1, 'same', bn.scale, bn.shift, ops.ACT_RELU6, out=y, partials=part))
    res.append('unfused fwd %.1f + %.1f' % (tu1, tu2))
    if ops.irb_supported((n, H, W, K), C, s, backward=True):
        for wa in ([int(os.environ['IRB_A_WAVES'])] if 'IRB_A_WAVES' in os.environ else [4096, 8192, 16384]):
            L.irb_set_bwd_plan(wa, 0)
            t = timed(lambda: ops.irb_bwd_sums(x, w1, bn, ops.ACT_RELU6, wdw, dy, s, in_scale=xs, in_shift=xh))
            res.append('passA w%d %.1f' % (wa, t))
        bn.coef.normal_()
        for wb in ([int(os.environ['IRB_B_WGS'])] if 'IRB_B_WGS' in os.environ else [512, 1024, 2048]):
            L.irb_set_bwd_plan(0, wb)
            t = timed(lambda: ops.irb_bwd_data(x, w1, bn, ops.ACT_RELU6, wdw, dy, s, in_scale=xs, in_shift=xh, out=gx,
                                               front=(x, xs, xh, ops.ACT_NONE, bn.mean[:K], bn.invstd[:K])))
            res.append('passB g%d %.1f' % (wb, t))
        L.irb_set_bwd_plan(0, 0)
    print(' | '.join(res), flush=True)
