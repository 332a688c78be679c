"""how much of a mid-size GEMM's time is tile quantisation?  one (K, N), the row count walked across the 256-CU boundaries
(16384 = 256 tiles of 64 rows, 17424 = the 33 x 33 x 16 maps = 273 tiles, ...), table off, a few pinned tiles.
GPU box: python3 scripts/micro/gemm_quant.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
dev = 'cuda'


def timeit(fn, reps=20):
    ts = []
    for i in range(reps + 5):
        L.probe_arm(3000 + i)
        fn()
    torch.cuda.synchronize()
    for i in range(5, reps + 5):
        ms = ctypes.c_float(0)
        L.probe_read(3000 + i, ctypes.addressof(ms))
        ts.append(ms.value)
    ts.sort()
    return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1)


_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
for (K, N) in [(576, 96), (960, 160), (160, 960), (384, 64), (320, 256), (96, 576)]:
    for M in (8192, 16384, 17424, 24576, 32768, 34848):
        x = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) / K ** 0.5
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
        part = ops.new_partials(N, dev)
        y = torch.empty(M, N, device=dev)
        fn = lambda: ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part)
        line = 'K=%4d N=%4d M=%6d (%4d tiles of 64)  table %6.1f us |' % (K, N, M, (M + 63) // 64, timeit(fn))
        L.set_option(b'gemm_tuned', 0)
        for nt, mi, pc in [(0, 0, 0), (min(8, (N + 15) // 16), 1, 0), (min(8, (N + 15) // 16), 1, 1), (min(8, (N + 15) // 16), 2, 0), (2, 1, 0), (4, 1, 0)]:
            L.set_option(b'gemm_nt', nt); L.set_option(b'gemm_mi', mi); L.set_option(b'gemm_per_cu', pc)
            line += ' [%d,%d,%d] %6.1f' % (nt, mi, pc, timeit(fn))
        L.set_option(b'gemm_nt', 0); L.set_option(b'gemm_mi', 0); L.set_option(b'gemm_per_cu', 0)
        L.set_option(b'gemm_tuned', 1)
        print(line + '   floor %.1f' % (2.0 * M * K * N / 155e6), flush=True)
