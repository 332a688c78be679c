"""random geometries through the split implicit-GEMM kernels of the dense convs (forward + statistics, data gradient, weight gradient on
every tile) against float64 torch: python3 scripts/micro/conv_sb_fuzz.py [cases] [seed]"""
import ctypes, importlib, os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L.set_option(b'conv_sb', 2)
done = worst = 0
while done < cases:
    k = int(rng.choice([1, 3, 3, 3, 5, 7]))
    s = int(rng.choice([1, 1, 2]))
    r = 1 if s == 2 else int(rng.choice([1, 1, 2, 3, 6]))
    Cin = 4 * int(rng.integers(1, 40))
    Cout = 4 * int(rng.integers(4, 40))
    H, W = int(rng.integers(9, 48)), int(rng.integers(9, 48))
    N = int(rng.integers(1, 4))
    pad = 'same'
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, k, s, r, pad)
    M, K = N * Ho * Wo, k * k * Cin
    if not (L.conv2d_gemm_supported(Cin, Cout, k, s) and L.conv2d_gemm_sb_supported(1, M, K, Cout) and L.conv2d_gemm_sb_supported(2, N * H * W, k * k * Cout, Cin)
            and L.conv2d_gemm_sb_supported(4, M, K, Cout)):
        continue
    g = torch.Generator(device=dev).manual_seed(done)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g)
    w = torch.randn(k, k, Cin, Cout, device=dev, generator=g) / K ** 0.5
    sc, sh = torch.rand(Cin, device=dev, generator=g) + 0.5, torch.randn(Cin, device=dev, generator=g) * 0.3
    gy = torch.randn(N, Ho, Wo, Cout, device=dev, generator=g)
    a64 = (x.double() * sc.double() + sh.double()).clamp(0, 6).permute(0, 3, 1, 2)
    ke = (k - 1) * r + 1
    th, tw = max((Ho - 1) * s + ke - H, 0), max((Wo - 1) * s + ke - W, 0)
    a64p = F.pad(a64, (pl, tw - pl, pt, th - pt)).requires_grad_(True)
    w64 = w.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    y64 = F.conv2d(a64p, w64, stride=s, dilation=r)
    y64.backward(gy.double().permute(0, 3, 1, 2))
    gx64 = a64p.grad[:, :, pt:pt + H, pl:pl + W].permute(0, 2, 3, 1)
    # the gradient w.r.t. the ACTIVATED input is what the data-gradient kernel produces (the activation's derivative is the caller's)
    gw64 = w64.grad.permute(2, 3, 1, 0)
    y64 = y64.detach().permute(0, 2, 3, 1)
    part = ops.new_partials(Cout, dev)
    y, rows = ops.conv2d_gemm_fwd_sb(x, w, s, r, pad, sc, sh, ops.ACT_RELU6, partials=part)
    p2 = part[:rows * 2 * Cout].reshape(rows, 2, Cout).double().sum(0)
    gx = ops.conv2d_gemm_bwd_data_sb(gy, w, (N, H, W, Cin), s, r, pad)
    errs = [float((y.double() - y64).abs().max() / y64.abs().max()), float((p2[0] - y64.reshape(-1, Cout).sum(0)).abs().max() / y64.abs().sum()) * Cout,
            float((gx.double() - gx64).abs().max() / gx64.abs().max())]
    for tile in (-1, 0, 1, 2, 3):
        L.set_option(b'split_wgrad_tile', tile)
        gw = ops.conv2d_gemm_bwd_weight(x, gy, k, s, r, pad, sc, sh, ops.ACT_RELU6)
        errs.append(float((gw.double() - gw64).abs().max() / gw64.abs().max()))
    L.set_option(b'split_wgrad_tile', -1)
    e = max(errs)
    worst = max(worst, e)
    flag = '' if e < 2e-5 else '   <-- LARGE'
    print('N=%d %2dx%-2d %3d->%3d k%d s%d r%d  M=%5d K=%4d | fwd %.1e stats %.1e dgrad %.1e wgrad %s%s' % (
        N, H, W, Cin, Cout, k, s, r, M, K, errs[0], errs[1], errs[2], ' '.join('%.1e' % v for v in errs[3:]), flag), flush=True)
    done += 1
print('worst %.2e over %d cases' % (worst, cases))
