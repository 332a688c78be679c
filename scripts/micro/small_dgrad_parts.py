"""the early-layer data gradients of the headline (129 x 129 / 257 x 257 maps, few channels: the streaming kernel pw_small_kernel) taken
apart: plain data gradient, with the fused BatchNorm-backward sums, and the forward of the mirrored shape with / without statistics --
time per launch (torch events around back-to-back launches) next to the HBM floor of what each one moves.
GPU box: python3 scripts/micro/small_dgrad_parts.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
dev = 'cuda'


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


# (rows, cin, cout) of the conv: the data gradient reduces over cout and writes cin columns
SHAPES = [(266256, 16, 96), (1065024, 32, 16), (266256, 96, 24), (266256, 24, 144), (266256, 144, 24), (66564, 144, 32)]
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=50)
del _w
st = torch.cuda.current_stream().cuda_stream
for (M, K, N) in SHAPES:
    dy = torch.randn(M, N, device=dev)
    w = torch.randn(K, N, device=dev) / N ** 0.5
    z = torch.randn(M, K, device=dev)
    gx = torch.empty(M, K, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    mu, inv = torch.randn(K, device=dev) * 0.2, torch.rand(K, device=dev) + 0.5
    part = ops.new_partials(max(K, N), dev)
    rows = ctypes.c_int(0)
    t0 = timeit(lambda: L.pwconv_bwd_data(dy.data_ptr(), N, w.data_ptr(), gx.data_ptr(), K, 0, M, K, N, st))
    t1 = timeit(lambda: L.pwconv_bwd_data_bn(dy.data_ptr(), N, w.data_ptr(), gx.data_ptr(), K, 0, M, K, N, z.data_ptr(), K, sc.data_ptr(),
                                               sh.data_ptr(), ops.ACT_RELU6, mu.data_ptr(), inv.data_ptr(), part.data_ptr(), ctypes.byref(rows), st))
    x = torch.randn(M, K, device=dev)
    wt = w.t().contiguous()
    y = torch.empty(M, N, device=dev)
    t2 = timeit(lambda: L.pwconv_fwd_wt(x.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, wt.data_ptr(), None, y.data_ptr(), N, None,
                                          ctypes.byref(rows), M, K, N, st))
    t3 = timeit(lambda: L.pwconv_fwd_wt(x.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, wt.data_ptr(), None, y.data_ptr(), N, part.data_ptr(),
                                          ctypes.byref(rows), M, K, N, st))
    fl = lambda nbytes: nbytes / 6.3e6
    print('M=%7d cin=%3d cout=%3d | dgrad %6.1f us (floor %5.1f)  + BN sums %6.1f (floor %5.1f) | fwd %6.1f  + stats %6.1f (floor %5.1f)' % (
        M, K, N, t0, fl(M * (K + N) * 4), t1, fl(M * (2 * K + N) * 4), t2, t3, fl(M * (K + N) * 4)), flush=True)
