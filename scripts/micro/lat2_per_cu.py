import ctypes, importlib, os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import torch
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
def timeit(fn, reps=20):
    ts = []
    for i in range(reps + 3):
        L.probe_arm(3000 + i); fn()
    torch.cuda.synchronize()
    for i in range(3, reps + 3):
        ms = ctypes.c_float(0); L.probe_read(3000 + i, ctypes.addressof(ms)); ts.append(ms.value)
    ts.sort(); return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1)
for (N, H, C, r) in [(4, 33, 2048, 18), (4, 33, 2048, 12), (16, 33, 320, 18), (2, 97, 2048, 36)]:
    x = torch.randn(N, H, H, C, device='cuda'); w = torch.randn(3, 3, C, device='cuda')
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda'); part = ops.new_partials(C, 'cuda'); y = torch.empty_like(x)
    by = 2 * N * H * H * C * 4
    line = 'N=%d %dx%dx%d rate %d (%.1f MB):' % (N, H, H, C, r, by / 1e6)
    for pc in (0, 1, 2, 3, 4, 6, 8):
        L.set_option(b'dw_per_cu', pc)
        t = timeit(lambda: ops.dwconv2d_fwd(x, w, 1, r, 'same', sc, sh, ops.ACT_RELU, out=y, partials=part))
        line += '  pc%d %.2f us (%.2f)' % (pc, t, by / t / 1e6 / 8)
    L.set_option(b'dw_per_cu', 0)
    print(line)
