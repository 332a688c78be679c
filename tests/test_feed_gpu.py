"""fit_generator's host -> device boundary (train.py:177-187): batches staged one ahead through pinned buffers and a copy stream
(model.BatchFeeder), the loss read one step late (model.LateScalar) -- the trajectory is the one train_on_batch gives when it is
handed the same host arrays one by one, bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu


class _Seq:
    def __init__(self, batches):
        self.b = batches
    def __len__(self):
        return len(self.b)
    def __getitem__(self, i):
        return self.b[i]


def _model(pkg, H, W, C, **kw):
    torch.manual_seed(0)
    m = pkg.get_deeplabv3p_model('mobilenetv2_lite', C, (H, W), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255), **kw)
    return m


@pytest.mark.parametrize('kind', ['uint8', 'float32', 'weights', 'adaptive'])
def test_staged_batches_give_the_trajectory_of_train_on_batch(kind):
    pkg = load_pkg()
    N, C, H, W = 2, 21, 65, 97
    rng = np.random.default_rng(5)
    batches = []
    for i in range(5):
        if kind in ('uint8', 'adaptive'):
            x = rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8)
            y = rng.integers(0, C + 3, (N, H * W, 1)).astype(np.uint8)
        else:
            x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
            y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
        if kind == 'weights':
            batches.append((x, y, {'pred_mask': rng.uniform(0.2, 3.0, (N, H * W)).astype(np.float32)}))
        else:
            batches.append((x, y))
    kw = dict(sample_weight_mode='temporal') if kind in ('weights', 'adaptive') else {}
    ref = _model(pkg, H, W, C, **kw)
    want = []
    for b in batches:
        sw = b[2]['pred_mask'] if len(b) > 2 else ('adaptive' if kind == 'adaptive' else None)
        want.append(ref.train_on_batch(b[0], b[1], sample_weight=sw))
    w_ref = ref.get_weights_by_name()
    m = _model(pkg, H, W, C, **kw)
    seen = []
    orig = m.train_on_batch
    m.train_on_batch = lambda *a, **k: (seen.append(type(a[0]).__name__), orig(*a, **k))[1]
    hist = m.fit_generator(_Seq(batches), steps_per_epoch=len(batches), epochs=1, verbose=0,
                           weighted_type='adaptive' if kind == 'adaptive' else None)
    assert seen == ['StagedBatch'] * len(batches)
    assert hist['loss'][0] == float(np.mean(want)), (hist['loss'][0], float(np.mean(want)))
    w = m.get_weights_by_name()
    assert all(np.array_equal(w[k], w_ref[k]) for k in w_ref)


def test_a_staging_slot_is_not_overwritten_before_its_step_has_read_it():
    """three batches in flight order with a slow consumer: slot 0 is reused by batch 2 only after step 0 has converted it"""
    pkg = load_pkg()
    N, C, H, W = 2, 21, 65, 65
    m = _model(pkg, H, W, C)
    rng = np.random.default_rng(1)
    xs = [rng.integers(0, 256, (N, H, W, 3)).astype(np.uint8) for _ in range(4)]
    ys = [rng.integers(0, C, (N, H * W, 1)).astype(np.uint8) for _ in range(4)]
    ref = _model(pkg, H, W, C)
    want = [ref.train_on_batch(x, y) for x, y in zip(xs, ys)]
    staged = m.prefetch_batch(xs[0], ys[0])
    got = []
    for i in range(4):
        t = m.train_on_batch(staged, None, return_tensor=True)
        if i + 1 < 4:
            staged = m.prefetch_batch(xs[i + 1], ys[i + 1])      # (lands in the slot step i - 1 used)
        got.append(float(t.item()))
    assert got == want, (got, want)


def test_nan_loss_stops_fit_one_step_late(monkeypatch):
    """TerminateOnNaN (train.py:64) with the loss read one step late: the NaN of step 1 is seen when step 2 has been enqueued"""
    pkg = load_pkg()
    model_mod = load_pkg('model')
    N, C, H, W = 2, 21, 65, 65
    m = _model(pkg, H, W, C)
    rng = np.random.default_rng(2)
    good = (rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32), rng.integers(0, C, (N, H * W, 1)).astype(np.float32))
    calls = []
    orig = m.train_on_batch
    m.train_on_batch = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    value = model_mod.LateScalar.value
    seen = []
    def poisoned(self, i):
        v = value(self, i)
        seen.append(v)
        return float('nan') if len(seen) == 2 else v          # the loss of step 1
    monkeypatch.setattr(model_mod.LateScalar, 'value', poisoned)
    m.fit_generator(_Seq([good] * 5), steps_per_epoch=5, epochs=3, verbose=0)
    assert m.stop_training and len(calls) == 3
