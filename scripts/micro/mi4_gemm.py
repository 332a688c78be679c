"""256-row tiles (gemm_mi = 4) of the fp32 tiled GEMM against 64- / 128-row tiles on the long GEMM shapes: forward (+ statistics)
and plain data gradient.  GPU box: python3 scripts/micro/mi4_gemm.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
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


for (M, K, N) in [(266256, 304, 256), (266256, 256, 256), (74498, 304, 256), (67600, 192, 64), (67600, 64, 384), (17424, 1280, 256),
                  (17424, 160, 960), (18818, 728, 728), (4356, 1536, 2048)]:
    x = torch.randn(M, K, device=dev)
    wt = torch.randn(N, K, device=dev) / K ** 0.5
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    part = ops.new_partials(N, dev)
    y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    w = wt.t().contiguous()
    gx = torch.empty(M, K, device=dev)
    line = 'M=%6d K=%4d N=%4d |' % (M, K, N)
    L.set_option(b'gemm_tuned', 1)
    t0 = timeit(lambda: ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))
    t1 = timeit(lambda: ops.pwconv_bwd_data(dy, w, out=gx))
    line += ' default fwd %6.1f dgrad %6.1f |' % (t0, t1)
    for nt in (8, 6, 4):
        for mi in (1, 2, 4):
            L.set_option(b'gemm_nt', nt); L.set_option(b'gemm_mi', mi)
            t0 = timeit(lambda: ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))
            t1 = timeit(lambda: ops.pwconv_bwd_data(dy, w, out=gx))
            line += ' [%d,%d] %6.1f %6.1f' % (nt, mi, t0, t1)
    L.set_option(b'gemm_nt', 0); L.set_option(b'gemm_mi', 0)
    print(line, flush=True)
