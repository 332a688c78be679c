"""The pinned-schedule data gradient with the folded BatchNorm-backward apply (pw_gemm_sb3d_kernel) against the row-stationary kernel,
through the library (event pair around the launch).  GPU box: python3 scripts/micro/sb3d_bench.py"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
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
    return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1)


_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=30)
del _w
COLD = os.environ.get('SB3_COLD', '1') == '1'
print('cold' if COLD else 'hot')
for M in (266256, 262144, 133128):
    K = N = 256
    g_ = torch.Generator(device=dev); g_.manual_seed(M)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g_)
    g, z_out, z = rnd(M, N), rnd(M, N), rnd(M, K)
    bsc, bsh = torch.rand(N, device=dev) + 0.5, rnd(N) * 0.3
    mu, istd = rnd(N) * 0.2, torch.rand(N, device=dev) + 0.5
    coef = torch.stack([torch.rand(N, device=dev) + 0.5, rnd(N) * 0.1, rnd(N) * 0.1]).contiguous()
    w_sp = ops.split_bf16x3((rnd(K, N) / 16).contiguous())
    sc, sh = torch.rand(K, device=dev) + 0.5, rnd(K) * 0.3
    mean, invstd = z.mean(0), 1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-3)
    part = ops.new_partials(K, dev)
    gx = torch.empty(M, K, device=dev)
    big = torch.empty(96 << 20, device=dev)
    row = '%7d x 256 -> 256' % M
    for name, sb3 in (('row-stationary', 0), ('pinned', 1)):
        L.set_option(b'sb3', sb3)
        for sums in (True, False):
            def run():
                if COLD:
                    big.zero_()
                gg = g          # in place: g is overwritten by dz (timing only)
                if sums:
                    ops.pwconv_bwd_data_sb_apply(gg, z_out, bsc, bsh, ops.ACT_RELU, mu, istd, coef, w_sp, N, out=gx, z=z, scale=sc, shift=sh,
                                                 act=ops.ACT_RELU, mean=mean, invstd=invstd, partials=part)
                else:
                    ops.pwconv_bwd_data_sb_apply(gg, z_out, bsc, bsh, ops.ACT_RELU, mu, istd, coef, w_sp, N, out=gx)
            row += '   %s%s %6.1f' % (name, '+sums' if sums else '', timeit(run))
    L.set_option(b'sb3', -1)
    print(row, flush=True)
    del g, z_out, z, gx, big
