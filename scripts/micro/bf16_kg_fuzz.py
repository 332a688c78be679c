"""random shapes through the bf16 tiled GEMM with its K groups pinned to 1 / 2 / 4 (forward + statistics, data gradient) against float64 on
bf16-rounded operands: python3 scripts/micro/bf16_kg_fuzz.py [cases] [seed]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for i in range(cases):
    M = int(rng.integers(65, 9000))
    K = 8 * int(rng.integers(4, 180))
    N = 8 * int(rng.integers(4, 100))
    g = torch.Generator(device=dev).manual_seed(i)
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    sc, sh = torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.3
    gy = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    wq = w.to(torch.bfloat16).double()
    a = (x.double() * sc.double() + sh.double()).float().clamp(0, 6).to(torch.bfloat16).double()        # fp32 prologue, rounded to bf16
    y64, gx64 = a @ wq, gy.double() @ wq.t()
    line = 'M=%5d K=%4d N=%4d |' % (M, K, N)
    ref = None
    for kg in (1, 2, 4):
        L.set_option(b'bf16_kg', kg)
        part = ops.new_partials(N, dev)
        y, rows = ops.pwconv_fwd_bf16(x, w, None, sc, sh, ops.ACT_RELU6, partials=part)
        gx = ops.pwconv_bwd_data_bf16(gy, w)
        p = part[:rows * 2 * N].reshape(rows, 2, N).double().sum(0)
        # in units of one bf16 ulp of the element (+ 1e-3 of the tensor's scale for sums that nearly cancel)
        ey = float(((y.double() - y64).abs() / (y64.abs() * 2.0 ** -8 + 1e-3 * y64.abs().max())).max())
        eg = float(((gx.double() - gx64).abs() / (gx64.abs() * 2.0 ** -8 + 1e-3 * gx64.abs().max())).max())
        es = float((p[0] - y.double().sum(0)).abs().max() / max(1e-30, float(y.double().abs().sum(0).max())))
        worst = max(worst, ey, eg)
        line += ' kg %d: y %.2f gx %.2f sums %.0e |' % (kg, ey, eg, es)
    print(line + ('   <-- LARGE' if max(ey, eg) > 1.0 else ''), flush=True)
L.set_option(b'bf16_kg', -1)
print('worst %.2f bf16 ulps over %d cases' % (worst, cases))
