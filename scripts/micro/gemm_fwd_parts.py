"""what the forward GEMM's prologue (producer BN + activation on the A tile) and BN-statistics epilogue cost:
python gemm_fwd_parts.py M K N"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
M, K, N = (int(a) for a in sys.argv[1:4])
NB = 4
xs = [torch.randn(M, K, device='cuda') for _ in range(NB)]
ys = [torch.empty(M, N, device='cuda') for _ in range(NB)]
w = torch.randn(K, N, device='cuda') / K ** 0.5
wt = w.t().contiguous()
sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda')
part = ops.new_partials(N, 'cuda')
i = [0]


def timeit(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(pro, stats):
    def f():
        i[0] = (i[0] + 1) % NB
        ops.pwconv_fwd_wt(xs[i[0]], wt, None, sc if pro else None, sh if pro else None, ops.ACT_RELU6 if pro else ops.ACT_NONE,
                          out=ys[i[0]], partials=part if stats else None)
    return timeit(f)


print('M=%d K=%d N=%d  plain %.1f us | +prologue %.1f | +stats %.1f | both %.1f'
      % (M, K, N, run(False, False), run(True, False), run(False, True), run(True, True)))
