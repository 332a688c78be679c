"""The fused inverted-residual blocks INSIDE the model (executor._find_irb): MobileNetV2's expand -> BatchNorm -> ReLU6 -> depthwise
runs of deeplabv3p_mobilenetv2.py:43-60 on csrc/irb_fwd.hip / irb_bwd.hip.  Production takes them from 131072 input pixels per batch
up (the 257 x 257 / 129 x 129 blocks of BASELINE configs[1]); DL3P_IRB_MIN_ROWS=1 brings the small parity models onto them."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
import test_model_gpu as TM

pytestmark = pytest.mark.gpu


@pytest.fixture
def fused(monkeypatch):
    monkeypatch.setenv('DL3P_IRB_MIN_ROWS', '1')
    monkeypatch.delenv('DL3P_IRB', raising=False)
    # the executor the bench runs: no debug copy of the expand output (the oracle tests rebuild the activation pattern of the expand
    # BatchNorm from the saved block input and the pre-step kernel: test_model_gpu._act_derivs)
    monkeypatch.delenv('DL3P_IRB_DEBUG_Z', raising=False)


def test_which_blocks_are_fused(fused, monkeypatch):
    pkg = load_pkg()
    monkeypatch.setenv('DL3P_IRB_DEBUG_Z', '1')
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    ex = m._executor(2, True)
    names = sorted(e.name for e in ex._irb_expand)
    # K in {16, 24, 32} with C = 6K: blocks 1 .. 6 (the first block has no expand conv, blocks 7+ have 64+ input channels)
    assert names == ['expanded_conv_%d_expand' % i for i in range(1, 7)], names
    for e in ex._irb_expand:
        assert e.out.id not in ex.grad          # the expanded tensor's gradient has no buffer ...
    monkeypatch.delenv('DL3P_IRB_DEBUG_Z')
    m1 = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    m1.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    ex1 = m1._executor(2, True)
    for e in ex1._irb_expand:
        assert e.out.id not in ex1.buf and e.out.id not in ex1.grad          # ... and outside the debug hook neither has the tensor
    # production threshold: nothing at this size
    monkeypatch.delenv('DL3P_IRB_MIN_ROWS')
    m2 = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    m2.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    assert not m2._executor(2, True)._irb_expand
    # a frozen backbone keeps the unfused kernels (the fused backward serves trainable blocks)
    monkeypatch.setenv('DL3P_IRB_MIN_ROWS', '1')
    m3 = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, freeze_level=1, training=True)
    m3.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    assert not m3._executor(2, True)._irb_expand
    # inference takes the fused forward
    m4 = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=False)
    assert len(m4._executor(2, False)._irb_expand) == 6


@pytest.mark.parametrize('model_type,H,W,OS', [('mobilenetv2', 65, 65, 16), ('mobilenetv2', 129, 97, 16), ('mobilenetv2_lite', 64, 80, 16),
                                               ('mobilenetv2', 65, 65, 8)])
def test_fused_blocks_do_not_change_the_step(model_type, H, W, OS, monkeypatch):
    """same weights, same batch: loss, every gradient, the updated weights and moving statistics with the fused blocks against the
    unfused kernels (the statistics of the expand BatchNorm come from the input covariance on one side, from the materialised tensor
    on the other)"""
    N, C = 2, 21
    x, y = TM._data(N, H, W, C, seed=17)

    def run(env):
        for k in ('DL3P_IRB', 'DL3P_IRB_MIN_ROWS'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(0)
        m, _ = TM._pair(model_type, H, W, C, OS=OS)
        m.use_graphs = False
        loss = m.train_on_batch(x, y)
        st = m._store
        ex = m._executor(N, True)
        g = {p.name: np.array(st.get(p, st.G), dtype=np.float64) for p in m.graph.all_params() if p.trainable}
        w = {k: np.array(v, dtype=np.float64) for k, v in m.get_weights_by_name().items()}
        return loss, g, w, len(ex._irb_expand), len(ex.fwd.items) + len(ex.bwd.items)

    l1, g1, w1, n1, k1 = run({'DL3P_IRB_MIN_ROWS': '1'})
    l0, g0, w0, n0, k0 = run({'DL3P_IRB': '0'})
    assert n1 >= 3 and n0 == 0
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l1, l0)
    # The fused forward rounds the expand conv differently (other k order) and takes the expand BatchNorm's statistics from the input
    # covariance: 1e-7 differences that, through 50 batch-statistics BatchNorms, flip a handful of ReLU6 branches downstream -- the
    # un-injected bound of tests/test_product_vs_transformers_gpu.py applies (the precise comparisons are the oracle tests below,
    # which inject the device's branch pattern, and tests/test_irb_gpu.py)
    num = sum(float(((a - g1[n]) ** 2).sum()) for n, a in g0.items())
    den = sum(float((a ** 2).sum()) for a in g0.values())
    assert np.sqrt(num / den) < 3e-2, np.sqrt(num / den)
    gmax = max(float(np.abs(a).max()) for a in g0.values())
    worst = ('', 0.0)
    for name, a in g0.items():
        r = float(np.abs(a - g1[name]).max() / (np.abs(a).max() + 1e-3 * gmax))
        if r > worst[1]:
            worst = (name, r)
    assert worst[1] < 0.35, worst
    for name, a in w0.items():
        lim = 1e-5 * max(1.0, float(np.abs(a).max())) + (0.01 * 0.35 * float(np.abs(g0[name]).max()) if name in g0 else 0.0)
        assert float(np.abs(a - w1[name]).max()) < lim, name


@pytest.mark.parametrize('graphs', [False, True])
def test_the_debug_copy_of_the_expand_output_changes_no_bit(graphs, monkeypatch):
    """DL3P_IRB_DEBUG_Z=1 (a test hook: the forward also writes the expand output with the unfused kernel) against the production plan
    on the same 129 x 129 MobileNetV2 step, every eligible block fused: loss, every gradient, every updated weight and moving
    statistic bit for bit -- the fused kernels never read the copy"""
    N, C, H, W = 2, 21, 129, 129
    x, y = TM._data(N, H, W, C, seed=23)
    monkeypatch.setenv('DL3P_IRB_MIN_ROWS', '1')
    monkeypatch.delenv('DL3P_IRB', raising=False)

    def run(hook):
        if hook:
            monkeypatch.setenv('DL3P_IRB_DEBUG_Z', '1')
        else:
            monkeypatch.delenv('DL3P_IRB_DEBUG_Z', raising=False)
        torch.manual_seed(0)
        m, _ = TM._pair('mobilenetv2', H, W, C)
        m.use_graphs = graphs
        losses = [m.train_on_batch(x, y) for _ in range(2)]
        st = m._store
        ex = m._executor(N, True)
        assert len(ex._irb_expand) == 6
        has_copy = [e.out.id in ex.buf for e in ex._irb_expand]
        g = {p.name: np.array(st.get(p, st.G)) for p in m.graph.all_params() if p.trainable}
        w = {k: np.array(v) for k, v in m.get_weights_by_name().items()}
        return losses, g, w, has_copy, len(ex.fwd.items)

    l1, g1, w1, c1, n1 = run(True)
    l0, g0, w0, c0, n0 = run(False)
    assert all(c1) and not any(c0) and n1 == n0 + 6
    assert l1 == l0, (l1, l0)
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), k
    for k in w0:
        assert np.array_equal(w0[k], w1[k]), k


def test_train_step_with_fused_blocks_matches_oracle(fused):
    TM.test_train_step_matches_oracle('mobilenetv2', 65, 65, 0, 16)


def test_train_step_with_fused_blocks_matches_oracle_os8(fused):
    TM.test_train_step_matches_oracle('mobilenetv2', 65, 65, 0, 8)


def test_predict_with_fused_blocks_matches_oracle(fused):
    TM.test_predict_matches_oracle('mobilenetv2', 65, 65)
    TM.test_predict_matches_oracle('mobilenetv2_lite', 65, 97)


def test_every_layer_with_fused_blocks_matches_float64(fused, monkeypatch):
    monkeypatch.setenv('DL3P_FOLD_APPLY', '0')
    monkeypatch.setenv('DL3P_GRAD_ALIAS', '0')
    TM._teacher_forced_step('mobilenetv2', 65, 65, 16, 2, 2e-5, 5e-4, 1e-3)


def test_graph_replay_equals_eager_with_fused_blocks(fused):
    TM.test_graph_replay_equals_eager()


def test_the_production_threshold_fuses_the_high_resolution_blocks():
    """at 513 x 513 the default rule takes the 257 x 257 block from batch 1 and the 129 x 129 blocks from batch 4 up"""
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 16, training=True)
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    assert sorted(e.name for e in m._executor(2, True)._irb_expand) == ['expanded_conv_1_expand']
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize('graphs', [False, True])
@pytest.mark.parametrize('sync_bn', [True, False])
def test_two_identical_ranks_reproduce_the_single_gpu_step_with_fused_blocks(graphs, sync_bn, monkeypatch):
    """data parallelism with the fused blocks (tests/test_dist_gpu.py's double: every all-reduce multiplies by the world size, which is
    exact in binary floating point): under SyncBatchNorm the expand BatchNorm's covariance sums and backward sums go through the
    staging buffer and the block's second pass is issued from the statistics flush -- the trajectory must be the single-GPU one, bit
    for bit"""
    import test_dist_gpu as TD
    monkeypatch.setenv('DL3P_IRB_MIN_ROWS', '1')
    monkeypatch.delenv('DL3P_IRB', raising=False)
    monkeypatch.delenv('DL3P_IRB_DEBUG_Z', raising=False)
    ctx = TD._two_identical_ranks()(sync_bn=sync_bn)
    monkeypatch.setenv('DL3P_FOLD_APPLY', '0')
    ref_l, ref_w, m0 = TD._trajectory('mobilenetv2', 65, 65, 2, None, 3, graphs)
    assert len(m0._executor(2, True)._irb_expand) == 6
    monkeypatch.delenv('DL3P_FOLD_APPLY')
    got_l, got_w, m = TD._trajectory('mobilenetv2', 65, 65, 2, ctx, 3, graphs)
    ex = m._executor(2, True)
    assert ex.dist is ctx and ex.sync_bn == sync_bn and len(ex._irb_expand) == 6
    assert got_l == ref_l, (got_l, ref_l)
    worst = max((float(np.abs(got_w[k] - ref_w[k]).max()), k) for k in ref_w)
    assert worst[0] == 0.0, worst
