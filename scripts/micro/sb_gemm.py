"""split-bf16 GEMM against the fp32 MFMA GEMM: time (library event pair) and error against float64 for the compute-bound
shapes of the BASELINE graphs.  GPU box: python3 scripts/micro/sb_gemm.py [nt mi]"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
if os.environ.get('DL3P_LIB_VARIANT'):          # A/B against a library built by build_variant.sh
    libm = importlib.import_module(PKG + '._lib')
    libm._lib = libm.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdl3p_%s.so' % os.environ['DL3P_LIB_VARIANT']))
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


SHAPES = [(266256, 304, 256), (266256, 256, 256), (17424, 1280, 256), (17424, 960, 160), (17424, 160, 960), (17424, 576, 96),
          (67600, 192, 64), (67600, 384, 64), (4356, 728, 728), (4356, 2048, 256), (4356, 1536, 2048), (18818, 728, 728), (74498, 304, 256)]
# (nt, mi, wm): wm = 0 -> the 2-workgroups-per-CU tiles (gemm_nt / gemm_mi), wm >= 1 with nt in (8, 12, 16) -> the wide family
# wm = -1 -> the producer / consumer form (sb_pipe) with (nt, mi)
if os.environ.get('SB_SHAPES'):          # e.g. SB_SHAPES=66564x256x256,66564x304x256
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
cands = [(None, None, None), (8, 2, -1), (8, 1, -1), (4, 2, -1), (4, 2, 0), (8, 2, 0), (16, 1, 2)]
if len(sys.argv) > 3:
    cands = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
elif len(sys.argv) > 1 and sys.argv[1] == 'tiles':
    cands = [(None, None, None), (8, 2, 0), (4, 2, 0), (16, 1, 2), (16, 2, 1)]
elif len(sys.argv) > 1 and sys.argv[1] == 'wide':
    cands = [(None, None, None), (16, 1, 2), (16, 1, 1), (12, 2, 1), (16, 2, 1), (8, 2, 2)]


def pin(nt, mi, wm):
    L.set_option(b'sb_pipe', 1 if wm == -1 else 0)
    if nt is None:
        for k in (b'gemm_nt', b'gemm_mi', b'sb_wm', b'sb_nt'):
            L.set_option(k, 0)
    elif wm == -1:
        L.set_option(b'sb_wm', 0); L.set_option(b'sb_nt', 0); L.set_option(b'gemm_nt', nt); L.set_option(b'gemm_mi', mi)
    elif wm == 0:
        L.set_option(b'sb_wm', -1); L.set_option(b'sb_nt', 0); L.set_option(b'gemm_nt', nt); L.set_option(b'gemm_mi', mi)
    else:
        L.set_option(b'sb_wm', wm); L.set_option(b'sb_nt', nt); L.set_option(b'gemm_mi', mi); L.set_option(b'gemm_nt', 0)
# the first timed launch of a process reads ~20 % high (clock ramp): burn one
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
del _w
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev)
    wt = torch.randn(N, K, device=dev) / K ** 0.5
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    part = ops.new_partials(N, dev)
    y = torch.empty(M, N, device=dev)
    wsp = ops.split_bf16x3(wt)
    y64 = (x.double() * sc.double() + sh.double()).clamp(0, 6) @ wt.double().t()
    t32 = timeit(lambda: ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))
    e32 = float((y.double() - y64).abs().max() / y64.abs().max())
    line = 'fwd  M=%6d K=%4d N=%4d  fp32 %7.1f us (err %.1e) |' % (M, K, N, t32, e32)
    for nt, mi, wm in cands:
        if wm and wm > 0 and nt and N <= 16 * (nt - 4):
            continue
        pin(nt, mi, wm)
        t = timeit(lambda: ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))
        e = float((y.double() - y64).abs().max() / y64.abs().max())
        line += ' [%s,%s,%s] %6.1f (%.0e)' % (nt, mi, wm, t, e)
    pin(None, None, None)
    print(line, flush=True)
    del y64
    # data gradient + BN sums: dy (M, N) -> gx (M, K)
    dy = torch.randn(M, N, device=dev)
    w = wt.t().contiguous()
    w_sp = ops.split_bf16x3(w)
    z = torch.randn(M, K, device=dev)
    mean, invstd = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    gx = torch.empty(M, K, device=dev)
    partk = ops.new_partials(K, dev)
    t32 = timeit(lambda: ops.pwconv_bwd_data_bn(dy, w, z, sc, sh, ops.ACT_RELU6, mean, invstd, partk, out=gx))
    line = 'dgbn M=%6d K=%4d N=%4d  fp32 %7.1f us             |' % (M, K, N, t32)
    for nt, mi, wm in cands:
        if wm and wm > 0 and nt and K <= 16 * (nt - 4):
            continue
        pin(nt, mi, wm)
        t = timeit(lambda: ops.pwconv_bwd_data_sb(dy, w_sp, N, out=gx, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean,
                                                  invstd=invstd, partials=partk))
        line += ' [%s,%s,%s] %6.1f        ' % (nt, mi, wm, t)
    pin(None, None, None)
    print(line, flush=True)
