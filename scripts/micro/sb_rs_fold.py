"""dev: the data gradient with the folded BatchNorm-backward apply (dl3p_pwconv_bwd_data_sb_apply) against bn_bwd_apply + the
row-stationary data gradient, decoder shapes.  GPU box: python3 scripts/micro/sb_rs_fold.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'


def timeit(fn, reps=10):
    ts = []
    for i in range(reps + 3):
        L.probe_arm(3000 + i)
        fn()
    torch.cuda.synchronize()
    for i in range(3, reps + 3):
        ms = ctypes.c_float(0)
        L.probe_read(3000 + i, ctypes.addressof(ms))
        ts.append(ms.value)
    ts.sort()
    return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1)


_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
for (M, K, N) in [(266256, 304, 256), (266256, 256, 256)]:
    g = torch.randn(M, N, device=dev); zo = torch.randn(M, N, device=dev)
    bsc, bsh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.3
    mu, istd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    coef = torch.stack([torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1]).contiguous()
    w = (torch.randn(K, N, device=dev) / N ** 0.5).contiguous(); w_sp = ops.split_bf16x3(w)
    z = torch.randn(M, K, device=dev); sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    mean, invstd = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    part = ops.new_partials(K, dev); gx = torch.empty(M, K, device=dev); dz = torch.empty(M, N, device=dev)
    t_fold = timeit(lambda: ops.pwconv_bwd_data_sb_apply(g, zo, bsc, bsh, ops.ACT_RELU6, mu, istd, coef, w_sp, N, dz=dz, out=gx, z=z, scale=sc,
                                                         shift=sh, act=ops.ACT_RELU6, mean=mean, invstd=invstd, partials=part))
    t_dg = timeit(lambda: ops.pwconv_bwd_data_sb(dz, w_sp, N, out=gx, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean, invstd=invstd, partials=part))
    ap = lambda: L.bn_bwd_apply(g.data_ptr(), N, zo.data_ptr(), N, bsc.data_ptr(), bsh.data_ptr(), ops.ACT_RELU6, mu.data_ptr(), istd.data_ptr(),
                                coef.data_ptr(), dz.data_ptr(), N, 0, M, N, None)
    # (bn_bwd_apply is not launched through the library's probe: torch events around 20 back-to-back launches)
    for _ in range(3):
        ap()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ap()
    e1.record(); torch.cuda.synchronize()
    t_ap = e0.elapsed_time(e1) * 1e3 / 20
    print('M=%d %d -> %d: folded %.1f us | apply %.1f + data gradient (+sums) %.1f = %.1f us' % (M, N, K, t_fold, t_ap, t_dg, t_ap + t_dg), flush=True)
