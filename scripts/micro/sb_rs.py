"""Row-stationary split-bf16 GEMM (csrc/pw_split_rs.hip, dl3p_set_option('sb_rs', 1)) against the tiled split kernels and float64:
forward + BatchNorm statistics, data gradient (+ fused BatchNorm-backward sums) on the decoder shapes of the BASELINE graphs.
GPU box: python3 scripts/micro/sb_rs.py [check]"""
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


def part_sums(part, rows, C):
    p = part[:rows * 2 * C].view(rows, 2, C).double().sum(0)
    return p[0], p[1]


# (M, K, N): forward K -> N; the data gradient of the same layer reduces over N and produces K columns
SHAPES = [(266256, 304, 256), (266256, 256, 256), (74498, 304, 256), (66564, 256, 256), (66564, 128, 128), (16641 * 2 + 7, 192, 48)]
if os.environ.get('SB_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
check_only = len(sys.argv) > 1 and sys.argv[1] == 'check'
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)      # clock ramp
del _w
for (M, K, N) in SHAPES:
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=dev)
    wt = torch.randn(N, K, device=dev) / K ** 0.5
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    bias = torch.randn(N, device=dev)
    wsp = ops.split_bf16x3(wt)
    a64 = (x.double() * sc.double() + sh.double()).clamp(0, 6)
    y64 = a64 @ wt.double().t() + bias.double()
    s64, q64 = y64.sum(0), (y64 * y64).sum(0)
    del a64
    res = {}
    for name, rs in (('tiled', 0), ('rs', 1)):
        L.set_option(b'sb_rs', rs)
        part = ops.new_partials(N, dev)
        y = torch.full((M, N), float('nan'), device=dev)
        _, rows = ops.pwconv_fwd_sb(x, wsp, K, bias, sc, sh, ops.ACT_RELU6, out=y, partials=part)
        torch.cuda.synchronize()
        e = float((y.double() - y64).abs().max() / y64.abs().max())
        s, q = part_sums(part, rows, N)
        es = float((s - s64).abs().max() / s64.abs().max())
        eq = float((q - q64).abs().max() / q64.abs().max())
        t = 0.0 if check_only else timeit(lambda: ops.pwconv_fwd_sb(x, wsp, K, bias, sc, sh, ops.ACT_RELU6, out=y, partials=part))
        res[name] = (t, e, es, eq, rows)
    print('fwd  M=%6d K=%4d N=%4d | ' % (M, K, N) + ' | '.join('%s %7.1f us err %.1e stats %.1e %.1e rows %d' % ((k,) + v) for k, v in res.items()), flush=True)
    del y64
    # data gradient + BatchNorm-backward sums: dy (M, N) -> gx (M, K), z (M, K)
    dy = torch.randn(M, N, device=dev)
    w = wt.t().contiguous()          # (K, N)
    w_sp = ops.split_bf16x3(w)
    z = torch.randn(M, K, device=dev)
    mean, invstd = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5
    g64 = dy.double() @ w.double().t()
    u = z.double() * sc.double() + sh.double()
    d64 = g64 * ((u > 0) & (u < 6)).double()
    s64, q64 = d64.sum(0), (d64 * (z.double() - mean.double()) * invstd.double()).sum(0)
    res = {}
    for name, rs in (('tiled', 0), ('rs', 1)):
        L.set_option(b'sb_rs', rs)
        partk = ops.new_partials(K, dev)
        gx = torch.full((M, K), float('nan'), device=dev)
        _, rows = ops.pwconv_bwd_data_sb(dy, w_sp, N, out=gx, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean, invstd=invstd, partials=partk)
        torch.cuda.synchronize()
        e = float((gx.double() - g64).abs().max() / g64.abs().max())
        s, q = part_sums(partk, rows, K)
        es = float((s - s64).abs().max() / s64.abs().max())
        eq = float((q - q64).abs().max() / q64.abs().max())
        t = 0.0 if check_only else timeit(lambda: ops.pwconv_bwd_data_sb(dy, w_sp, N, out=gx, z=z, scale=sc, shift=sh, act=ops.ACT_RELU6, mean=mean, invstd=invstd, partials=partk))
        gx2 = torch.full((M, K), float('nan'), device=dev)
        ops.pwconv_bwd_data_sb(dy, w_sp, N, out=gx2)          # plain data gradient
        torch.cuda.synchronize()
        e2 = float((gx2.double() - g64).abs().max() / g64.abs().max())
        t2 = 0.0 if check_only else timeit(lambda: ops.pwconv_bwd_data_sb(dy, w_sp, N, out=gx2))
        res[name] = (t, e, es, eq, t2, e2)
    print('dgbn M=%6d K=%4d N=%4d | ' % (M, K, N) + ' | '.join('%s %7.1f us err %.1e sums %.1e %.1e ; plain %7.1f us err %.1e' % ((k,) + v) for k, v in res.items()), flush=True)
    L.set_option(b'sb_rs', -1)
    del g64, d64, u
