"""what does the vendor library (MIOpen through torch.nn.functional.conv2d, groups = C) need for the depthwise convolutions of
the headline step?  A yardstick for the `roofline` object, not a dependency.  GPU box: python3 scripts/micro/vendor_dw.py"""
import ctypes, importlib, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'


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


for (N, H, W, C, k, stride, rate, name) in [(16, 33, 33, 320, 3, 1, 18, 'aspp3_depthwise (the roofline kernel)'), (16, 33, 33, 320, 3, 1, 12, 'aspp2_depthwise'),
                                            (16, 33, 33, 320, 3, 1, 6, 'aspp1_depthwise'), (16, 129, 129, 304, 3, 1, 1, 'decoder_conv0_depthwise'),
                                            (16, 257, 257, 96, 3, 2, 1, 'expanded_conv_1_depthwise'), (16, 33, 33, 960, 3, 1, 2, 'expanded_conv_14_depthwise')]:
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(k, k, C, device=dev)
    y = ops.dwconv2d_fwd(x, w, stride, rate)
    t_ours = lib_time(lambda: ops.dwconv2d_fwd(x, w, stride, rate, out=y))
    xt = x.permute(0, 3, 1, 2)                                   # NCHW view of NHWC memory = channels_last
    wt = w.permute(2, 0, 1).unsqueeze(1).contiguous()            # (C, 1, k, k)
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, k, stride, rate, 'same')
    pad_b = max((Ho - 1) * stride + (k - 1) * rate + 1 - H - pt, 0)
    pad_r = max((Wo - 1) * stride + (k - 1) * rate + 1 - W - pl, 0)

    def vendor(xin):
        xp = F.pad(xin, (pl, pad_r, pt, pad_b)) if (pt != pad_b or pl != pad_r) else xin
        return F.conv2d(xp, wt, None, stride, (0, 0) if xp is not xin else (pt, pl), rate, C)
    ref = vendor(xt)
    err = float((ref.permute(0, 2, 3, 1) - y).abs().max())
    t_cl = ev_time(lambda: vendor(xt))
    xc = xt.contiguous()                                         # plain NCHW
    t_nchw = ev_time(lambda: vendor(xc))
    nbytes = (x.numel() + y.numel()) * 4
    print('%-40s N=%d %dx%dx%d k=%d s=%d r=%d  MIOpen channels_last %7.1f us  NCHW %7.1f us | dl3p %6.1f us (%.2f TB/s)  max|diff| %.1e'
          % (name, N, H, W, C, k, stride, rate, t_cl, t_nchw, t_ours, nbytes / t_ours / 1e6, err), flush=True)
