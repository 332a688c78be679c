"""staged fit_generator step: which ingredient costs (side stream copy, graphs, events)"""
import importlib, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
N, C, H, W = 16, 21, 513, 513
rng = np.random.default_rng(0)
b8 = [(rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8), rng.integers(0, C, (N, H * W, 1)).astype(np.uint8)) for _ in range(3)]


def run(tag, graphs=True, side=True, K=20):
    m = pkg.get_deeplabv3p_model('mobilenetv2', C, (H, W), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    m.use_graphs = graphs
    f = m._feeder()
    if not side:
        f.stream = torch.cuda.current_stream()
    staged = m.prefetch_batch(*b8[0])
    for i in range(5):
        m.train_on_batch(staged, None)
        staged = m.prefetch_batch(*b8[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        lt = m.train_on_batch(staged, None, return_tensor=True)
        staged = m.prefetch_batch(*b8[(i + 1) % 3])
        float(lt.item())
    torch.cuda.synchronize()
    print('%-40s %.2f ms/step' % (tag, 1e3 * (time.perf_counter() - t0) / K), flush=True)
    del m
    torch.cuda.empty_cache()


run('graphs, side-stream copy')
run('graphs, copy on the compute stream', side=False)
run('eager, side-stream copy', graphs=False)
run('eager, copy on the compute stream', graphs=False, side=False)
