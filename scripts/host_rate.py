"""images/s of train_on_batch when every batch arrives as host numpy arrays (the reference's fit_generator boundary):
PCIe copy of 16 x 513 x 513 x 3 floats + labels per step included -- reported in DESIGN.md, never as bench `value`"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
N, C, H, W = 16, 21, 513, 513
m = pkg.get_deeplabv3p_model('mobilenetv2', C, (H, W), 16, training=True)
m.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
rng = np.random.default_rng(0)
batches = [(rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32), rng.integers(0, C, (N, H * W, 1)).astype(np.float32))
           for _ in range(3)]
for i in range(4):
    m.train_on_batch(*batches[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for i in range(K):
    m.train_on_batch(*batches[i % 3])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('train_on_batch with host numpy batches: %.1f images/s (%.2f ms/step)' % (N * K / dt, 1e3 * dt / K))

# the same with the bytes the generator decoded: uint8 pixels / uint8 labels, normalised on the device
batches8 = [(rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8), rng.integers(0, C, (N, H * W, 1)).astype(np.uint8))
            for _ in range(3)]
for i in range(4):
    m.train_on_batch(*batches8[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    m.train_on_batch(*batches8[i % 3])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('train_on_batch with host uint8 batches:  %.1f images/s (%.2f ms/step)' % (N * K / dt, 1e3 * dt / K))


# fit_generator (train.py:177-187): the same uint8 batches through a Sequence -- batch k + 1 is copied into pinned memory and crosses
# PCIe on the copy stream while step k runs, the loss is read one step late (model.BatchFeeder / LateScalar)
class _Seq:
    def __init__(self, batches, n):
        self.b, self.n = batches, n
    def __len__(self):
        return self.n
    def __getitem__(self, i):
        return self.b[i % len(self.b)]
for name, bb in (('uint8', batches8), ('float32', batches)):
    m.fit_generator(_Seq(bb, 6), steps_per_epoch=6, epochs=1, verbose=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fit_generator(_Seq(bb, 3 * K), steps_per_epoch=3 * K, epochs=1, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('fit_generator with host %-7s batches (staged ahead): %.1f images/s (%.2f ms/step)' % (name, N * 3 * K / dt, 1e3 * dt / (3 * K)))
# ... and the resident rate of the same executor (inputs left where they are): what bench.py reports as `value`
ex = m._executor(N, True)
for i in range(5):
    ex.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3 * K):
    ex.train_step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('resident inputs (bench.py):                              %.1f images/s (%.2f ms/step)' % (N * 3 * K / dt, 1e3 * dt / (3 * K)))


# --weighted_type adaptive (deeplabv3p/data.py:134-145): the generator's per-image balanced class weights.  Host side as
# the reference computes them (np.unique + one putmask per class; sklearn's formula inlined) against 'adaptive' on the
# device (dl3p_label_prepare)
m2 = pkg.get_deeplabv3p_model('mobilenetv2', C, (H, W), 16, training=True)
m2.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255),
           sample_weight_mode='temporal')


def host_weights(lab_u8):
    out = np.zeros((N, H * W), dtype='float32')
    for n in range(N):
        label = lab_u8[n].astype('int32').flatten()
        label[label > (C - 1)] = 255
        class_list, counts = np.unique(label, return_counts=True)
        cw = label.size / (len(class_list) * counts.astype(np.float64))
        for class_id, w in zip(class_list, cw):
            np.putmask(out[n], label == class_id, w)
    return out


t0 = time.perf_counter()
sw = host_weights(batches8[0][1])
t_host = time.perf_counter() - t0
for mode in ('host', 'device'):
    for i in range(3):
        m2.train_on_batch(*batches8[i % 3], sample_weight='adaptive')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        x8, y8 = batches8[i % 3]
        if mode == 'host':
            m2.train_on_batch(x8, y8, sample_weight=host_weights(y8))
        else:
            m2.train_on_batch(x8, y8, sample_weight='adaptive')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('adaptive pixel weights on the %-6s: %.1f images/s (%.2f ms/step; host weights alone %.1f ms per batch)'
          % (mode, N * K / dt, 1e3 * dt / K, 1e3 * t_host))
