"""The pinned-schedule split forward (csrc/pw_split3.hip) against the tiled split kernels on the long decoder shapes, through the
library (event pair around the launch).  GPU box: python3 scripts/micro/sb3_bench.py [relu-fraction]"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
L.set_option(b'pw_small_min_rows', -1)
dev = 'cuda'


def timeit(fn, reps=12):
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
    return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1), 1e3 * ts[-1]


_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
del _w
SHAPES = [(266256, 304, 256, 304), (266256, 304, 256, 320), (262144, 304, 256, 304), (262144, 304, 256, 320), (262144, 320, 256, 320), (262144, 256, 256, 256)]
COLD = os.environ.get('SB3_COLD', '1') == '1' 
if os.environ.get('SB_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
print('cold' if COLD else 'hot')
for (M, K, N, ldx) in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(M + K)
    xb = torch.randn(M, ldx, device=dev, generator=g)
    x = xb[:, :K]
    wt = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    sc = torch.rand(K, device=dev, generator=g) + 0.5
    sh = torch.randn(K, device=dev, generator=g) * 0.3
    wsp = ops.split_bf16x3(wt)
    y = torch.empty(M, N, device=dev)
    part = ops.new_partials(N, dev)
    # something else between the launches, as in the step: the operand is not in the caches when the GEMM starts
    big = torch.empty(96 << 20, device=dev)
    row = '%7d x %4d (ld %3d) -> %3d' % (M, K, ldx, N)
    for name, sb3 in (('tiled', 0), ('pinned', 1)):
        L.set_option(b'sb3', sb3)
        for stats in (True, False):
            def run():
                if COLD:
                    big.zero_()
                ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU, out=y, partials=part if stats else None)
            def arm_run():
                run()
            # (probe_arm arms the NEXT GEMM launch: the zero_() in front is a torch kernel and does not take it)
            t, worst = timeit(run)
            row += '   %s%s %6.1f (max %6.1f)' % (name, '+stats' if stats else '', t, worst)
    L.set_option(b'sb3', -1)
    print(row, flush=True)
    del x, xb, y, big
