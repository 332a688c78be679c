"""dev instrument: where a pointwise-GEMM wave spends its cycles (needs scripts/micro/build_stamp.sh)
phases per wave: 0 stage (vmcnt wait + LDS writes)  1 barrier  2 prefetch issue + fragment reads + MFMAs
                 3 barrier  4 epilogue"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
libm = importlib.import_module('tf-keras-deeplabv3p-model-set_amd._lib')
libm._lib = libm.Lib(os.path.join(ROOT, 'scripts/micro/libdl3p_stamp.so'))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
shapes = [(266256, 304, 256), (266256, 256, 256), (4356, 728, 728), (17424, 960, 160), (17424, 160, 960), (17424, 1280, 256)]
dbg = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device='cuda')
os.environ['DL3P_STAMP_PTR'] = str(dbg.data_ptr())
for M, K, N in shapes:
    x = torch.randn((M, K), device='cuda'); w = torch.randn((K, N), device='cuda') * 0.05
    sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda') * 0.1
    y = torch.empty((M, N), device='cuda'); dy = torch.randn((M, N), device='cuda'); gx = torch.empty((M, K), device='cuda')
    part = ops.new_partials(N, 'cuda')
    for name, f in (('fwd', lambda: ops.pwconv_fwd(x, w, None, sc, sh, ops.ACT_RELU6, out=y, partials=part)),
                    ('dgrad', lambda: ops.pwconv_bwd_data(dy, w, out=gx))):
        for _ in range(2):
            dbg.zero_()
            f()
        torch.cuda.synchronize()
        d = dbg.view(-1, 8).cpu().numpy()
        d = d[d[:, 5] > 0]
        its = d[:, 5].mean()
        ph = d[:, :5].mean(0)
        tot = ph.sum()
        fmt = 'stage %6.0f  bar1 %6.0f  mfma %6.0f  bar2 %6.0f  epi %6.0f'
        print(('M=%7d K=%4d N=%4d %-5s waves %5d  k-steps/wave %5.1f  total %8.0f cyc | per k-step: ' + fmt)
              % (M, K, N, name, len(d), its, tot, *(ph / its)), flush=True)
