"""column-mean bias of the split-bf16 GEMMs: worst |mean(y - y64)| / sigma(y64) over the output columns, beside the rms error per
element, for the fp32-input MFMA kernel and the split kernel (forward role, inputs behind a BN + ReLU prologue).
GPU box: python3 scripts/micro/sb_bias.py   (SB_SHAPES=MxKxN,...)"""
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
dev = 'cuda'
SHAPES = [(66564, 304, 256), (66564, 256, 256), (17424, 960, 320), (17424, 1280, 256), (17424, 320, 1280)]
if os.environ.get('SB_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
L.set_option(b'pw_small_min_rows', 1 << 30)
for (M, K, N) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, N, device=dev, generator=g) / K ** 0.5
    sc, sh = torch.rand(K, device=dev, generator=g) + 0.5, torch.randn(K, device=dev, generator=g) * 0.3
    a64 = (x.double() * sc.double() + sh.double()).clamp(min=0)
    y64 = a64 @ w.double()
    sig = y64.std(0)
    wt = w.t().contiguous()
    wsp = ops.split_bf16x3(wt)
    line = 'M=%6d K=%4d N=%4d |' % (M, K, N)
    y0 = ops.pwconv_fwd_wt(x, wt, None, sc, sh, ops.ACT_RELU)
    for name, y in (('fp32', y0), ('split', ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU))):
        e = y.double() - y64
        line += ' %s: mean bias %.1e sigma, rms %.1e |' % (name, float((e.mean(0).abs() / sig).max()), float((e ** 2).mean().sqrt() / (y64 ** 2).mean().sqrt()))
    print(line, flush=True)
