"""dev: s_memtime stamps of workgroup 0 of the row-stationary split GEMM (ablation build, DL3P_SB_ABLATE=100): per slot and wave,
cycles from the slot's head to (1) the multiply loop, (2) its end, (3) the barrier.
GPU box: DL3P_SB_ABLATE=100 DL3P_LIB_VARIANT=abl python3 scripts/micro/sb_rs_stamps.py [M K N]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
libm = importlib.import_module(PKG + '._lib')
libm._lib = libm.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdl3p_%s.so' % os.environ.get('DL3P_LIB_VARIANT', 'abl')))
ops = importlib.import_module(PKG + '.ops')
L = libm.lib()
L.set_option(b'pw_small_min_rows', -1)
L.set_option(b'sb_rs', 1)
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (266256, 256, 256)
x = torch.randn(M, K, device='cuda')
wt = torch.randn(N, K, device='cuda') / K ** 0.5
sc, sh = torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda') * 0.3
wsp = ops.split_bf16x3(wt)
part = torch.zeros(2048 * 2 * N + 8 * 128 * 8 * 2 + 64, dtype=torch.float32, device='cuda')
y = torch.empty(M, N, device='cuda')
for _ in range(20):
    part[2048 * 2 * N:].zero_()
    ops.pwconv_fwd_sb(x, wsp, K, None, sc, sh, ops.ACT_RELU6, out=y, partials=part)
torch.cuda.synchronize()
st = part[2048 * 2 * N: 2048 * 2 * N + 8 * 128 * 8 * 2].cpu().numpy().view(np.uint64).reshape(8, 128, 8).astype(np.int64)
t0 = st[:, 0, 0].min()
for w in (0, 4):
    print('wave', w)
    for s in range(0, 40):
        a = st[w, s]
        if a[0] == 0:
            continue
        nxt = st[w, s + 1, 0] if s + 1 < 128 and st[w, s + 1, 0] else 0
        if a[4]:
            print('  slot %2d head +%7d | STAGING: lazy stats %5d | chunks' % (s, a[0] - t0, a[1] - a[0]), [int(a[4] - a[1])] + [int(a[5 + i] - a[4 + i]) for i in range(3)], '| rest %5d | barrier->next head %5d' % (a[3] - a[7], (nxt - a[3]) if nxt else -1))
            continue
        print('  slot %2d head +%7d | to loop %5d | loop %5d | epilogue %5d | barrier->next head %5d' % (
            s, a[0] - t0, (a[1] - a[0]) if a[1] else -1, (a[2] - a[1]) if a[2] else -1, (a[3] - (a[2] if a[2] else a[0])), (nxt - a[3]) if nxt else -1))
