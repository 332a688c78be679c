"""dl3p_pwconv_fwd_wt against dl3p_pwconv_fwd_wt_splitk on the ASPP forward shapes of configs[2] (4356 x {2048, 1280} -> 256, with the
BatchNorm statistic rows): us per launch over slices / tile knobs (dl3p_set_option("splitk", S), DL3P_SPLITK_NT / _MI)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
L = ops.lib()
dev = 'cuda'
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
shapes = [(4356, 2048, 256), (4356, 1280, 256), (18818, 2048, 256), (17424, 2048, 256), (17424, 1280, 256)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (M, K, N) in shapes:
    xs = [torch.randn(M, K, device=dev) for _ in range(3)]
    wt = torch.randn(N, K, device=dev) / K ** 0.5
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    y = torch.empty(M, N, device=dev)
    part = ops.new_partials(N, dev)
    i = [0]
    def one():
        i[0] = (i[0] + 1) % 3
        ops.pwconv_fwd_wt(xs[i[0]], wt, None, sc, sh, ops.ACT_RELU6, out=y, partials=part)
    line = 'M=%d K=%d N=%d: one launch %.1f us |' % (M, K, N, timeit(one))
    for S in (2, 4, 5, 8, 10, 16):
        L.set_option(b'splitk', S)
        if L.pwconv_fwd_splitk_plan(M, K, N) != S:
            continue
        wsb = L.pwconv_fwd_splitk_workspace(M, K, N)
        ws = torch.empty(wsb // 4, device=dev)
        rows = __import__('ctypes').c_int(0)
        def sk():
            i[0] = (i[0] + 1) % 3
            L.pwconv_fwd_wt_splitk(xs[i[0]].data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, wt.data_ptr(), None, y.data_ptr(), N,
                                   part.data_ptr(), __import__('ctypes').byref(rows), ws.data_ptr(), wsb, M, K, N, None)
        line += ' S=%d %.1f' % (S, timeit(sk))
    L.set_option(b'splitk', -1)
    print(line, '(rule: S=%d)' % L.pwconv_fwd_splitk_plan(M, K, N), flush=True)
