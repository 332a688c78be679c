"""instrumented build (build_stamp.sh): how much of the tiled GEMM's time is the B-operand staging?
DL3P_GEMM_STAGGER = 0 (normal) | 103 (no B LDS stores after the first K-step, loads not waited for) | 104 (no B loads
either) | 105 (loads waited for, not stored); results are
wrong by construction in the ablation modes -- only the time matters.  usage: python ablate_b.py"""
import os, sys, importlib, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    libm = importlib.import_module('tf-keras-deeplabv3p-model-set_amd._lib')
    libm._lib = libm.Lib(os.path.join(ROOT, 'scripts/micro/libdl3p_stamp.so'))
    ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
    for M, K, N in [(266256, 304, 256), (266256, 256, 256), (4356, 728, 728), (17424, 960, 160)]:
        dy = torch.randn((M, N), device='cuda'); w = torch.randn((K, N), device='cuda') * 0.05
        gx = torch.empty((M, K), device='cuda')
        f = lambda: ops.pwconv_bwd_data(dy, w, out=gx)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        print('  dgrad M=%d K=%d N=%d: %.1f us' % (M, K, N, e0.elapsed_time(e1) * 100))
else:
    for mode in ('0', '100', '102', '103', '105', '106'):
        print('DL3P_GEMM_STAGGER=' + mode, flush=True)
        env = dict(os.environ, DL3P_GEMM_STAGGER=mode)
        subprocess.run([sys.executable, __file__, 'child'], env=env)
