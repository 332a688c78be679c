"""split-bf16 weight gradient (pw_wgrad_sb_kernel) against the fp32-input MFMA weight gradient: kernel time (library event pair, slabs
left for the batched reduction) and error against float64.  GPU box: python3 scripts/micro/sb_wgrad.py   (SB_SHAPES=MxKxN,...)"""
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


SHAPES = [(266256, 304, 256), (266256, 256, 256), (17424, 1280, 256), (17424, 960, 160), (17424, 160, 960), (17424, 960, 320), (17424, 320, 256),
          (17424, 576, 96), (17424, 384, 64), (67600, 192, 64), (67600, 144, 32), (4356, 728, 728), (66564, 256, 256), (18818, 728, 728)]
if os.environ.get('SB_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)      # clock ramp
del _w
ws = torch.empty(96 << 20, device=dev)      # 384 MB of slab space
st = torch.cuda.current_stream().cuda_stream
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev)
    dy = torch.randn(M, N, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    a64 = (x.double() * sc.double() + sh.double()).clamp(0, 6)
    gw64 = a64.t() @ dy.double()
    del a64
    rows = ctypes.c_int(0)

    def run():
        L.pwconv_bwd_weight_slabs(x.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, dy.data_ptr(), N, ws.data_ptr(),
                                  ws.numel() * 4, ctypes.byref(rows), M, K, N, st)
    line = 'wgrad M=%6d K=%4d N=%4d |' % (M, K, N)
    for v in (0, 1):
        L.set_option(b'split_wgrad', v)
        t = timeit(run)
        gw = ws[:rows.value * K * N].reshape(rows.value, K, N).double().sum(0)
        e = float((gw - gw64).abs().max() / gw64.abs().max())
        # the reduction of the slabs at the rate the batched launch streams them (tune_gemm.py)
        line += ' %s %7.1f us + %5.1f (%3d slabs, err %.1e) |' % ('split' if v else 'fp32 ', t, rows.value * K * N * 4 / 3.3e6, rows.value, e)
    L.set_option(b'split_wgrad', 0)
    print(line, flush=True)
