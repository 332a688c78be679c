"""what does the vendor library reach on the decoder GEMM shapes?  torch.mm (rocBLAS / hipBLASLt, fp32 in, fp32 out, plain GEMM:
no prologue, no statistics) next to dl3p_pwconv_fwd_wt without prologue / statistics.  A yardstick for roofline_mfma, not a
dependency: the product never calls it.  GPU box: python3 scripts/micro/vendor_gemm.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
dev = 'cuda'
torch.backends.cuda.matmul.allow_tf32 = False


def ev_time(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return sum(ts[:reps // 2]) / (reps // 2)


def lib_time(fn, reps=20):
    ts = []
    for i in range(reps + 5):
        L.probe_arm(3000 + i)
        fn()
    torch.cuda.synchronize()
    for i in range(5, reps + 5):
        ms = ctypes.c_float(0)
        L.probe_read(3000 + i, ctypes.addressof(ms))
        ts.append(ms.value * 1e3)
    ts.sort()
    return sum(ts[:reps // 2]) / (reps // 2)


_w = torch.randn(65536, 256, device=dev)
lib_time(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
SHAPES = [(266256, 304, 256), (266256, 256, 256), (17424, 960, 160), (17424, 320, 256), (17424, 1280, 256), (4356, 728, 728), (18818, 1536, 1536)]
if len(sys.argv) > 3:
    SHAPES = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev)
    wt = torch.randn(N, K, device=dev) / K ** 0.5
    w = wt.t().contiguous()
    y = torch.empty(M, N, device=dev)
    t_nt = ev_time(lambda: torch.mm(x, wt.t(), out=y))
    t_nn = ev_time(lambda: torch.mm(x, w, out=y))
    t_ours_ev = ev_time(lambda: ops.pwconv_fwd_wt(x, wt, out=y))
    t_ours = lib_time(lambda: ops.pwconv_fwd_wt(x, wt, out=y))
    fl = 2.0 * M * K * N
    print('M=%6d K=%4d N=%4d  torch.mm NT %7.1f us (%5.1f TF)  NN %7.1f us (%5.1f TF) | dl3p %7.1f us kernel (%5.1f TF), %7.1f us launch-to-done'
          % (M, K, N, t_nt, fl / t_nt / 1e6, t_nn, fl / t_nn / 1e6, t_ours, fl / t_ours / 1e6, t_ours_ev), flush=True)
