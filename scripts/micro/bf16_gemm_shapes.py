"""the bf16 pointwise GEMMs of configs[4] (MobileNetV3-Large, 1024 x 2048, one image per device) with few rows and a long reduction --
time per launch (torch events around back-to-back launches; weights pre-converted) and error against float64.
GPU box: python3 scripts/micro/bf16_gemm_shapes.py     (BF_SHAPES=MxKxN,...; DL3P_BF16_KG to pin the K-group count)"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'


def timeit(fn, reps=50):
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


# forward shapes (rows, cin, cout) of the 64 x 128 and 128 x 256 maps
SHAPES = [(8192, 80, 480), (8192, 480, 112), (8192, 112, 672), (8192, 672, 112), (8192, 672, 160), (8192, 160, 960), (8192, 960, 160),
          (8192, 160, 256), (8192, 1280, 256), (8192, 200, 80), (8192, 80, 200), (32768, 40, 120), (32768, 120, 40), (32768, 72, 40),
          (131072, 304, 256), (131072, 256, 256)]
if os.environ.get('BF_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['BF_SHAPES'].split(',')]
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=50)      # clock ramp
del _w
st = torch.cuda.current_stream().cuda_stream
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(K, N, device=dev) / K ** 0.5
    wt = w.t().contiguous().to(torch.bfloat16)
    wb = w.contiguous().to(torch.bfloat16)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    gy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    gx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    part = ops.new_partials(max(N, K), dev)
    rows = ctypes.c_int(0)
    a64 = (x.double() * sc.double() + sh.double()).clamp(min=0).to(torch.bfloat16).double()
    y64 = a64 @ wt.double().t()
    gx64 = gy.double() @ wb.double().t()
    f = lambda: L.pwconv_fwd_bf16(x.data_ptr(), K, 0, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU, wt.data_ptr(), None, y.data_ptr(), N, 0,
                                  part.data_ptr(), ctypes.byref(rows), M, K, N, st)
    tf = timeit(f)
    ef = float((y.double() - y64).abs().max() / y64.abs().max())
    p2 = part[:rows.value * 2 * N].reshape(rows.value, 2, N).double().sum(0)
    es = float((p2[0] - y.double().sum(0)).abs().max() / y.double().abs().sum(0).max())
    fr = rows.value
    f2 = lambda: L.pwconv_fwd_bf16(x.data_ptr(), K, 0, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU, wt.data_ptr(), None, y.data_ptr(), N, 0,
                                   None, ctypes.byref(rows), M, K, N, st)
    f3 = lambda: L.pwconv_fwd_bf16(x.data_ptr(), K, 0, None, None, ops.ACT_NONE, wt.data_ptr(), None, y.data_ptr(), N, 0,
                                   None, ctypes.byref(rows), M, K, N, st)
    tf2, tf3 = (timeit(f2), timeit(f3)) if os.environ.get('BF_PARTS') else (0.0, 0.0)
    d = lambda: L.pwconv_bwd_data_bf16(gy.data_ptr(), N, 0, wb.data_ptr(), gx.data_ptr(), K, 0, M, K, N, st)
    td = timeit(d)
    ed = float((gx.double() - gx64).abs().max() / gx64.abs().max())
    byt = (M * K + M * N) * 2
    print('M=%6d K=%4d N=%4d | fwd+stats %6.1f us (err %.1e, sums %.1e, %3d rows; no stats %.1f, no prologue either %.1f) | dgrad %6.1f us (err %.1e) | HBM floor %4.1f us' % (
        M, K, N, tf, ef, es, fr, tf2, tf3, td, ed, byt / 6.3e6), flush=True)
