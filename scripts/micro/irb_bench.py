"""fused inverted-residual kernels (csrc/irb_fwd.hip, irb_bwd.hip) at the BASELINE configs[1] launch shapes, next to the unfused
kernels they replace; raw C-ABI calls on preallocated buffers (kernel time, no wrapper work).
knobs: IRB_CT (channel tiles per wave of the forward), IRB_WAVES, IRB_A_WAVES, IRB_B_WAVES, IRB_N"""
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


def knob(name, default):
    return [int(os.environ[name])] if name in os.environ else default


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
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, 3, s, 1, 'same')
    dy = torch.randn((n, Ho, Wo, C), device='cuda')
    part = ops.new_partials(C, 'cuda')
    y = torch.empty((n, Ho, Wo, C), device='cuda')
    gx = torch.empty((n, H, W, K), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    cov_rows = torch.empty(256 * (K + K * K), dtype=torch.float64, device='cuda')
    cov_sums = torch.empty(K + K * K, dtype=torch.float64, device='cuda')
    rows = ctypes.c_int(0)
    geo = (n, H, W, K, C, s, pt, pl, Ho, Wo)
    M = n * H * W

    def stats():
        L.irb_cov_stats(P(x), K, P(xs), P(xh), 0, P(cov_rows), ctypes.byref(rows), M, K, st)
        L.irb_cov_reduce(P(cov_rows), rows.value, K, P(cov_sums), st)
        L.irb_bn_finalize_cov(P(cov_sums), P(w1), K, C, float(M), P(bn.gamma), P(bn.beta), bn.eps, bn.momentum, P(bn.moving_mean),
                              P(bn.moving_var), 1, P(bn.scale), P(bn.shift), P(bn.mean), P(bn.invstd), st)
    t_stats = timed(stats)
    t_cov = timed(lambda: L.irb_cov_stats(P(x), K, P(xs), P(xh), 0, P(cov_rows), ctypes.byref(rows), M, K, st))
    head = (P(x), K, P(xs), P(xh), 0, P(w1), P(bn.scale), P(bn.shift), ops.ACT_RELU6)
    res = ['%dx%dx%dx%d->%d s%d' % (n, H, W, K, C, s), 'cov %.1f +reduce+finalize %.1f' % (t_cov, t_stats)]
    for ct in knob('IRB_CT', [1, 2]):
        for waves in knob('IRB_WAVES', [2048, 4096, 8192]):
            L.irb_set_plan(ct, waves)
            t = timed(lambda: L.irb_fwd(*head, P(wdw), P(y), C, P(part), ctypes.byref(rows), *geo, st))
            res.append('fwd ct%d w%d %.1f' % (ct, waves, t))
    L.irb_set_plan(0, 0)
    # the unfused pair
    z1 = torch.empty((n, H, W, C), device='cuda')
    p2 = ops.new_partials(C, 'cuda')
    tu1 = timed(lambda: L.pwconv_fwd_wt(P(x), K, P(xs), P(xh), 0, P(w1t), None, P(z1), C, P(p2), ctypes.byref(rows), M, K, C, st))
    tu2 = timed(lambda: L.dwconv2d_fwd(P(z1), C, P(bn.scale), P(bn.shift), ops.ACT_RELU6, P(wdw), P(y), C, P(part), ctypes.byref(rows),
                                       n, H, W, C, 3, s, 1, pt, pl, Ho, Wo, st))
    res.append('unfused fwd %.1f + %.1f' % (tu1, tu2))
    if ops.irb_supported((n, H, W, K), C, s, backward=True):
        bn.coef.normal_()
        bn.mean.normal_(); bn.invstd.uniform_(0.5, 2.0)
        for wa in knob('IRB_A_WAVES', [4096, 8192, 16384]):
            L.irb_set_bwd_plan(wa, 0)
            nb = L.irb_bwd_workspace(0, n, H, W, K, C, s, pt, pl)
            slabs = torch.empty(nb // 4, device='cuda')
            t = timed(lambda: L.irb_bwd_sums(*head, P(bn.mean), P(bn.invstd), P(wdw), P(dy), C, P(slabs), nb, ctypes.byref(rows), P(part), *geo, st))
            res.append('passA w%d %.1f' % (wa, t))
        part0 = ops.new_partials(K, 'cuda')
        for wb in knob('IRB_B_WAVES', [3072, 6144, 12288]):
            L.irb_set_bwd_plan(0, wb)
            nb = L.irb_bwd_workspace(1, n, H, W, K, C, s, pt, pl)
            slabs = torch.empty(nb // 4, device='cuda')
            t = timed(lambda: L.irb_bwd_data(*head, P(bn.mean), P(bn.invstd), P(bn.coef), P(wdw), P(dy), C, P(slabs), nb, ctypes.byref(rows),
                                             P(gx), K, 0, P(x), K, P(xs), P(xh), 0, P(bn.mean), P(bn.invstd), P(part0), *geo, st))
            res.append('passB w%d %.1f' % (wb, t))
        L.irb_set_bwd_plan(0, 0)
    print(' | '.join(res), flush=True)
