"""The split-bf16 weight gradient per tile (pw_wgrad_sb_kernel: index 0..4 = 128x128, 64x128, 128x64, 64x64, 128x256) and workgroups per
CU, pinned through dl3p_set_option("split_wgrad_tile" / "split_wgrad_per_cu"): kernel time and error against float64.
GPU box: SB_SHAPES=MxKxN,... python3 scripts/micro/sb_wgrad_tiles.py"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
PKG = 'tf-keras-deeplabv3p-model-set_amd'
ops = importlib.import_module(PKG + '.ops')
L = importlib.import_module(PKG + '._lib').lib()
dev = 'cuda'


def timeit(fn, reps=10):
    ts = []
    for i in range(reps + 3):
        L.probe_arm(3000 + i)
        fn()
    torch.cuda.synchronize()
    for i in range(3, reps + 3):
        ms = ctypes.c_float(0)
        L.probe_read(3000 + i, ctypes.addressof(ms))
        ts.append(ms.value)
    ts.sort()
    return 1e3 * sum(ts[:reps // 2 + 1]) / (reps // 2 + 1)


SHAPES = [(266256, 304, 256), (266256, 256, 256), (17424, 1280, 256), (17424, 960, 320), (17424, 320, 256), (66564, 256, 256), (18818, 728, 728)]
if os.environ.get('SB_SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['SB_SHAPES'].split(',')]
ws = torch.empty(96 << 20, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.set_option(b'split_wgrad', 1)
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev)
    dy = torch.randn(M, N, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    gw64 = (x.double() * sc.double() + sh.double()).clamp(0, 6).t() @ dy.double()
    rows = ctypes.c_int(0)

    def run():
        L.pwconv_bwd_weight_slabs(x.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ops.ACT_RELU6, dy.data_ptr(), N, ws.data_ptr(),
                                  ws.numel() * 4, ctypes.byref(rows), M, K, N, st)
    line = 'wgrad M=%6d K=%4d N=%4d |' % (M, K, N)
    combos = [(-1, 0), (4, 2), (4, 3)] if os.environ.get('SB_QUICK') else [(-1, 0), (0, 2), (0, 3), (4, 1), (4, 2), (4, 3)]
    for tile, pc in combos:
        L.set_option(b'split_wgrad_tile', tile); L.set_option(b'split_wgrad_per_cu', pc)
        t = timeit(run)
        gw = ws[:rows.value * K * N].reshape(rows.value, K, N).double().sum(0)
        e = float((gw - gw64).abs().max() / gw64.abs().max())
        line += ' [%d,%d] %6.1f (%3d slabs, %.0e) |' % (tile, pc, t, rows.value, e)
    L.set_option(b'split_wgrad_tile', -1); L.set_option(b'split_wgrad_per_cu', 0)
    print(line, flush=True)
