"""bf16 depthwise launches of BASELINE configs[4] (MobileNetV3-Large 1024 x 2048, batch 1) and of the 513 x 513 batch-16 graphs:
forward (+ prologue + statistics), data gradient, weight gradient, event-timed.  DL3P_BF16_DW_WINDOW=0 | 1 picks the strip kernels of
dw_bf16_strip.h or the sliding-window kernels of dwconv.hip (read once per process).  GPU box: python3 scripts/micro/bf16_dw.py"""
import importlib, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
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


SHAPES = [(1, 256, 512, 304, 3, 1, 1), (1, 256, 512, 256, 3, 1, 1), (1, 512, 1024, 64, 3, 2, 1), (1, 256, 512, 72, 5, 2, 1), (1, 64, 128, 960, 5, 1, 2),
          (1, 128, 256, 120, 5, 1, 1), (1, 256, 512, 72, 3, 1, 1), (1, 64, 128, 672, 5, 1, 1), (1, 512, 1024, 16, 3, 1, 1), (1, 64, 128, 672, 3, 1, 1),
          (1, 128, 256, 240, 3, 2, 1), (1, 64, 128, 160, 3, 1, 18), (16, 129, 129, 304, 3, 1, 1), (16, 33, 33, 960, 3, 1, 2)]
if len(sys.argv) > 1 and sys.argv[1] == 'xception':
    SHAPES = [(4, 33, 33, 728, 3, 1, 1), (4, 33, 33, 1024, 3, 1, 1), (4, 33, 33, 1536, 3, 1, 2), (4, 33, 33, 2048, 3, 1, 6), (4, 65, 65, 256, 3, 1, 1), (4, 65, 65, 728, 3, 2, 1),
              (4, 129, 129, 128, 3, 1, 1), (4, 129, 129, 256, 3, 2, 1), (4, 257, 257, 64, 3, 1, 1), (4, 257, 257, 128, 3, 2, 1), (4, 129, 129, 304, 3, 1, 1),
              (8, 129, 129, 64, 3, 1, 1)]
print('window kernels:', os.environ.get('DL3P_BF16_DW_WINDOW', '1'))
for (N, H, W, C, k, s, r) in SHAPES:
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    w = torch.randn(k, k, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    part = ops.new_partials(C, dev)
    y, _ = ops.dwconv2d_fwd_bf16(x, w, s, r, 'same', sc, sh, ops.ACT_RELU6, partials=part)
    dy = torch.randn_like(y.float()).to(torch.bfloat16)
    gx = torch.empty_like(x)
    tf = ev_time(lambda: ops.dwconv2d_fwd_bf16(x, w, s, r, 'same', sc, sh, ops.ACT_RELU6, partials=part))
    td = ev_time(lambda: ops.dwconv2d_bwd_data_bf16(dy, w, tuple(x.shape), s, r, out=gx))
    tw = ev_time(lambda: ops.dwconv2d_bwd_weight_bf16(x, dy, k, s, r, 'same', sc, sh, ops.ACT_RELU6))
    by = (x.numel() + y.numel()) * 2
    print('N=%2d %4dx%-4d C=%4d k=%d s=%d r=%-2d  fwd %7.1f us (%.2f TB/s)  dgrad %7.1f (%.2f)  wgrad+reduce %7.1f (%.2f)'
          % (N, H, W, C, k, s, r, tf, by / tf / 1e6, td, by / td / 1e6, tw, by / tw / 1e6), flush=True)
