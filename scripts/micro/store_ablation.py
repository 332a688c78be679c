"""What the write-heavy forward kernels of the headline cost without their stores (variant library built with
-DDL3P_ABLATE_STORES: scripts/micro/build_variant.sh nost -DDL3P_ABLATE_STORES; DL3P_LIB_OVERRIDE selects it).  Timing only."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
dev = 'cuda'


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


N = 16
# stem: 513 x 513 x 3 -> 257 x 257 x 32, stride 2
x = torch.rand(N, 513, 513, 3, device=dev) * 2 - 1
w = torch.randn(3, 3, 3, 32, device=dev)
part = ops.new_partials(32, dev)
print('stem_conv_fwd            %6.1f us' % timeit(lambda: ops.stem_conv_fwd(x, w, partials=part)), flush=True)
for (M, K, Nn) in [(N * 129 * 129, 24, 144), (N * 257 * 257, 32, 16), (N * 129 * 129, 144, 24), (N * 129 * 129, 96, 24)]:
    a = torch.randn(M, K, device=dev)
    wt = torch.randn(Nn, K, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    y = torch.empty(M, Nn, device=dev)
    part = ops.new_partials(Nn, dev)
    print('pwconv_fwd_wt %7d x %3d -> %3d  %6.1f us' % (M, K, Nn, timeit(lambda: ops.pwconv_fwd_wt(a, wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part))), flush=True)
