import os, sys, ctypes, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
libm = importlib.import_module('tf-keras-deeplabv3p-model-set_amd._lib')
libm._lib = libm.Lib(os.path.join(ROOT, 'scripts/micro/libdl3p_stamp.so'))
ops = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.ops')
M, K, N = 266256, 256, 256
dbg = torch.zeros(4096 * 16, dtype=torch.int64, device='cuda')
os.environ['DL3P_STAMP_PTR'] = str(dbg.data_ptr())
w = torch.randn((K, N), device='cuda') * 0.05
dy = torch.randn((M, N), device='cuda'); gx = torch.empty((M, K), device='cuda')
for _ in range(3):
    ops.pwconv_bwd_data(dy, w, out=gx)
torch.cuda.synchronize()
d = dbg.view(-1, 8).cpu().numpy()[:, :5]
d = d[d.sum(1) > 0]
print('waves', len(d), 'mean cycles per wave: stage %.0f  barrier1 %.0f  mfma-phase %.0f  barrier2 %.0f epilogue(+vmcnt0) %.0f' % tuple(d.mean(0)))
tot = d.sum(1).mean()
print('shares: stage %.1f%% b1 %.1f%% mfma %.1f%% b2 %.1f%% epi %.1f%% total %.0f' % (*(100 * d.mean(0) / tot), tot))
