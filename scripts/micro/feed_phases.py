"""where the time of a staged fit_generator step goes on the host (model.BatchFeeder): per-phase wall clock"""
import importlib, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
N, C, H, W = 16, 21, 513, 513
m = pkg.get_deeplabv3p_model('mobilenetv2', C, (H, W), 16, training=True)
m.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
rng = np.random.default_rng(0)
b8 = [(rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8), rng.integers(0, C, (N, H * W, 1)).astype(np.uint8)) for _ in range(3)]
for i in range(4):
    m.train_on_batch(*b8[i % 3])
torch.cuda.synchronize()
print('torch threads', torch.get_num_threads())
f = m._feeder()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
staged = m.prefetch_batch(*b8[0])
m.train_on_batch(staged, None)
staged = m.prefetch_batch(*b8[1])
m.train_on_batch(staged, None)
staged = m.prefetch_batch(*b8[2])
K = 20
t_all = time.perf_counter()
for i in range(K):
    t0 = time.perf_counter(); lt = m.train_on_batch(staged, None, return_tensor=True); tick('enqueue step', t0)
    x, y = b8[(i + 1) % 3]
    slot = f.k & 1
    t0 = time.perf_counter(); f.ready[slot].synchronize(); tick('ready.sync', t0)
    t0 = time.perf_counter(); tx = torch.as_tensor(x).reshape(-1); ty = torch.as_tensor(y).reshape(-1); tick('as_tensor', t0)
    hx, dx = f.bufs[slot]['x']; hy, dy = f.bufs[slot]['y']
    t0 = time.perf_counter(); hx.copy_(tx); hy.copy_(ty); tick('host->pinned', t0)
    t0 = time.perf_counter(); staged = m.prefetch_batch(x, y); tick('prefetch_batch (all)', t0)
    t0 = time.perf_counter(); v = float(lt.item()); tick('loss.item', t0)
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print('%.2f ms/step' % (1e3 * dt / K))
for k, v in T.items():
    print('  %-24s %.2f ms/step' % (k, 1e3 * v / K))
import numpy
t0 = time.perf_counter()
for i in range(10):
    numpy.copyto(hx.numpy(), b8[i % 3][0].reshape(-1))
print('numpy.copyto into pinned: %.2f ms' % (1e2 * (time.perf_counter() - t0)))
torch.set_num_threads(8)
t0 = time.perf_counter()
for i in range(10):
    hx.copy_(torch.as_tensor(b8[i % 3][0]).reshape(-1))
print('torch copy_, 8 threads: %.2f ms' % (1e2 * (time.perf_counter() - t0)))
