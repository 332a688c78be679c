import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29512')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
sync_bn = os.environ.get('SYNC_BN', '1') == '1'
m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 16)
m.compile(optimizer=pkg.SGD(0.01), sync_bn=sync_bn)
ex = m._executor(16, True)
ex.train_step()
in_graph = os.environ.get('IN_GRAPH', '1') == '1'
for pl in (ex.fwd, ex.bwd, ex.opt):
    pl.capture(collectives_in_graph=in_graph)
print('segments fwd/bwd/opt', len(ex.fwd.segments), len(ex.bwd.segments), len(ex.opt.segments))
def t(f, R=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / R * 1e3
print('fwd %.2f ms  bwd %.2f ms  opt %.2f ms  step %.2f ms' % (t(ex.fwd.run), t(ex.bwd.run), t(ex.opt.run), t(ex.train_step)))
# raw tiny all-reduce cost
s = torch.zeros(640, dtype=torch.float64, device='cuda')
print('eager all_reduce(5KB) %.1f us' % (t(lambda: dist.all_reduce(s), 200) * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    for _ in range(100): dist.all_reduce(s)
print('in-graph all_reduce(5KB) %.1f us each' % (t(g.replay, 20) * 1e3 / 100))
dist.destroy_process_group()
