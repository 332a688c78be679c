"""dense k x k convolutions as implicit GEMMs: the split-bf16 kernels (dl3p_conv2d_gemm_fwd_sb / _bwd_data_sb, the split route of
dl3p_conv2d_gemm_bwd_weight_slabs) against the fp32-input MFMA kernels -- time per launch (torch events around 20 back-to-back launches)
and error against float64 torch.  GPU box: python3 scripts/micro/conv_sb.py   (CONV_SHAPES=NxHxWxCinxCoutxkxsxr,...)"""
import ctypes, importlib, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'


def timeit(fn, reps=20):
    for _ in range(3):
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


# (N, H, W, Cin, Cout, k, stride, rate)
SHAPES = [(4, 257, 257, 32, 64, 3, 1, 1),            # Xception entry_flow_conv1_2, configs[2] per device
          (2, 385, 385, 32, 64, 3, 1, 1),            # configs[3]
          (8, 129, 129, 64, 64, 3, 1, 1),            # ResNet50 stage 2 at 513^2, batch 8
          (8, 129, 129, 128, 128, 3, 2, 1),          # stage 3's strided 3x3
          (8, 65, 65, 128, 128, 3, 1, 1),
          (8, 65, 65, 256, 256, 3, 2, 1),
          (8, 33, 33, 256, 256, 3, 1, 1),
          (8, 33, 33, 512, 512, 3, 1, 2)]            # stage 5 at output stride 16
if os.environ.get('CONV_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['CONV_SHAPES'].split(',')]
_w = torch.randn(65536, 256, device=dev)
timeit(lambda: ops.pwconv_fwd_wt(_w, _w[:256].contiguous()), reps=50)      # clock ramp
del _w
ws = torch.empty(96 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream
for (N, H, W, Cin, Cout, k, s, r) in SHAPES:
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, k, s, r, 'same')
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
    gy = torch.randn(N, Ho, Wo, Cout, device=dev)
    K = k * k * Cin
    wt = w.reshape(K, Cout).t().contiguous()
    wsp = ops.split_bf16x3(wt)
    wd = torch.empty(Cin, k * k * Cout, device=dev)
    L.conv2d_gemm_dgrad_weights(w.data_ptr(), wd.data_ptr(), k, Cin, Cout, st)
    wdsp = ops.split_bf16x3(wd)
    y = torch.empty(N, Ho, Wo, Cout, device=dev)
    gx = torch.empty(N, H, W, Cin, device=dev)
    part = ops.new_partials(Cout, dev)
    rows = ctypes.c_int(0)
    # float64 references on the device (pad as TF 'same' does: explicit)
    a64 = (x.double() * sc.double() + sh.double()).clamp(min=0).permute(0, 3, 1, 2)
    ke = (k - 1) * r + 1
    tot_h, tot_w = max((Ho - 1) * s + ke - H, 0), max((Wo - 1) * s + ke - W, 0)
    a64p = F.pad(a64, (pl, tot_w - pl, pt, tot_h - pt)).requires_grad_(True)
    w64 = w.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    y64 = F.conv2d(a64p, w64, stride=s, dilation=r)
    y64.backward(gy.double().permute(0, 3, 1, 2))
    gx64 = a64p.grad[:, :, pt:pt + H, pl:pl + W].permute(0, 2, 3, 1)
    gw64 = w64.grad.permute(2, 3, 1, 0).reshape(K, Cout)
    y64 = y64.detach().permute(0, 2, 3, 1)
    geo = (N, H, W, Cin, Cout, k, s, r, pt, pl, Ho, Wo)

    def err(a, b):
        if os.environ.get('CONV_RMS'):
            return float(((a.double() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
        return float((a.double() - b).abs().max() / b.abs().max())
    M = N * Ho * Wo
    line = 'N=%d %dx%d %d->%d k%d s%d r%d |' % (N, H, W, Cin, Cout, k, s, r)
    f0 = lambda: L.conv2d_gemm_fwd(x.data_ptr(), Cin, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU, wt.data_ptr(), None, y.data_ptr(), Cout,
                                   part.data_ptr(), ctypes.byref(rows), *geo, st)
    f1 = lambda: L.conv2d_gemm_fwd_sb(x.data_ptr(), Cin, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU, wsp.data_ptr(), wsp.shape[2], None,
                                      y.data_ptr(), Cout, part.data_ptr(), ctypes.byref(rows), *geo, st)
    def stat_err():
        p2 = part[:rows.value * 2 * Cout].reshape(rows.value, 2, Cout).double().sum(0)
        r1, r2 = y64.reshape(-1, Cout).sum(0), (y64 ** 2).reshape(-1, Cout).sum(0)
        mean, var = p2[0] / M, p2[1] / M - (p2[0] / M) ** 2
        mr, vr = r1 / M, r2 / M - (r1 / M) ** 2
        return float(((mean - mr).abs() / vr.sqrt()).max()), float(((var - vr).abs() / vr).max())
    t0 = timeit(f0); e0 = err(y, y64); s0 = stat_err()
    L.set_option(b'conv_sb', 2)
    t1 = timeit(f1); e1 = err(y, y64); s1 = stat_err()
    line += ' fwd(M=%d K=%d) fp32 %6.1f us (%.0e; mean %.1e var %.1e) split %6.1f (%.0e; mean %.1e var %.1e) |' % ((M, K, t0, e0) + s0 + (t1, e1) + s1)
    d0 = lambda: L.conv2d_gemm_bwd_data(gy.data_ptr(), Cout, wd.data_ptr(), gx.data_ptr(), Cin, 0, *geo, st)
    d1 = lambda: L.conv2d_gemm_bwd_data_sb(gy.data_ptr(), Cout, wdsp.data_ptr(), wdsp.shape[2], gx.data_ptr(), Cin, 0, *geo, st)
    t0 = timeit(d0); e0 = err(gx, gx64)
    t1 = timeit(d1); e1 = err(gx, gx64)
    line += ' dgrad fp32 %6.1f (%.0e) split %6.1f (%.0e) |' % (t0, e0, t1, e1)

    def wg():
        L.conv2d_gemm_bwd_weight_slabs(x.data_ptr(), Cin, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU, gy.data_ptr(), Cout, ws.data_ptr(),
                                       ws.numel() * 4, ctypes.byref(rows), *geo, st)
    for v in (0, 2):
        L.set_option(b'conv_sb', v)
        t = timeit(wg)
        gw = ws[:rows.value * K * Cout].reshape(rows.value, K, Cout).double().sum(0)
        line += ' wgrad %s %6.1f (%d slabs, %.0e)' % ('split' if v else 'fp32', t, rows.value, err(gw, gw64))
    L.set_option(b'conv_sb', -1)
    print(line, flush=True)
