"""The round-5 shortcuts INSIDE the model: the split-K forward of the few-row / long-reduction 1x1 convs (Xception's ASPP,
dl3p_pwconv_fwd_wt_splitk), the stem's BatchNorm-backward apply inside its weight gradient (dl3p_stem_conv_bwd_weight_slabs_bn) and the
separable training head (dl3p_head_train_rows) are what a default executor traces, and switching each off leaves the step where it was
(same weights, same batch; loss, every gradient, the updated weights) -- at the bound of two summation orders, not bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
import test_model_gpu as TM

pytestmark = pytest.mark.gpu


def _step(model_type, H, W, C, N, monkeypatch, env, OS=16, options=None):
    import importlib
    pkg = load_pkg()
    L = importlib.import_module(pkg.__name__ + '.ops').lib()
    for k in ('DL3P_FOLD_APPLY_STEM', 'DL3P_FUSED_HEAD'):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for k, v in (options or {}).items():
        L.set_option(k, v)
    try:
        torch.manual_seed(0)
        x, y = TM._data(N, H, W, C, seed=23)
        m, _ = TM._pair(model_type, H, W, C, OS=OS)
        m.use_graphs = False
        loss = m.train_on_batch(x, y)
        st = m._store
        ex = m._executor(N, True)
        calls = [ep for plan in (ex.fwd, ex.bwd) for (ep, _) in plan.labels]
        g = {p.name: np.array(st.get(p, st.G), dtype=np.float64) for p in m.graph.all_params() if p.trainable}
        w = {k: np.array(v, dtype=np.float64) for k, v in m.get_weights_by_name().items()}
        return loss, g, w, calls
    finally:
        for k in (options or {}):
            L.set_option(k, -1)


def _same_step(a, b, rel_l2, worst_rel):
    (l1, g1, w1, _), (l0, g0, w0, _) = a, b
    assert abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    num = sum(float(((v - g1[n]) ** 2).sum()) for n, v in g0.items())
    den = sum(float((v ** 2).sum()) for v in g0.values())
    assert np.sqrt(num / den) < rel_l2, np.sqrt(num / den)
    gmax = max(float(np.abs(v).max()) for v in g0.values())
    for name, v in g0.items():
        r = float(np.abs(v - g1[name]).max() / (np.abs(v).max() + 1e-3 * gmax))
        assert r < worst_rel, (name, r)
    for name, v in w0.items():
        assert float(np.abs(v - w1[name]).max()) <= 1e-5 * max(1.0, float(np.abs(v).max())) + 0.01 * worst_rel * (
            float(np.abs(g0[name]).max()) if name in g0 else 0.0), name


def test_xception_aspp_forwards_take_the_split_k_form_and_the_step_stays(monkeypatch):
    # 4 x 257 x 257, OS 16: the ASPP map is 17 x 17, 1156 rows x 2048 -> 256: the rule serves it (few rows, K >= 1024)
    on = _step('xception', 257, 257, 21, 4, monkeypatch, {})
    off = _step('xception', 257, 257, 21, 4, monkeypatch, {}, options={b'splitk': 0})
    assert on[3].count('dl3p_pwconv_fwd_wt_splitk') >= 4, [c for c in on[3] if 'pwconv_fwd' in c][:12]
    assert 'dl3p_pwconv_fwd_wt_splitk' not in off[3]
    # another association of a 2048-term fp32 sum in five layers; a handful of ReLU branches may flip downstream
    _same_step(on, off, 3e-2, 0.35)


def test_stem_fold_and_separable_head_are_the_default_and_the_step_stays(monkeypatch):
    on = _step('mobilenetv2', 129, 129, 21, 2, monkeypatch, {})
    off = _step('mobilenetv2', 129, 129, 21, 2, monkeypatch, {'DL3P_FOLD_APPLY_STEM': '0', 'DL3P_FUSED_HEAD': '0'})
    assert 'dl3p_stem_conv_bwd_weight_slabs_bn' in on[3] and 'dl3p_head_train_rows' in on[3]
    assert 'dl3p_stem_conv_bwd_weight_slabs_bn' not in off[3] and 'dl3p_head_train_rows' not in off[3]
    assert 'dl3p_upsample_softmax_loss' in off[3] and 'dl3p_resize_bilinear_bwd' in off[3]
    _same_step(on, off, 3e-2, 0.35)


def test_batched_wgrad_off_at_a_fused_shape_keeps_the_unfused_blocks(monkeypatch):
    """the fused inverted-residual backward leaves its weight gradients as slabs for the batched reduction, so with
    DL3P_BATCHED_WGRAD=0 (an A/B switch) a training executor must not fuse -- it used to fuse and die on an assert while tracing"""
    monkeypatch.setenv('DL3P_IRB_MIN_ROWS', '1')
    monkeypatch.delenv('DL3P_IRB', raising=False)
    on = _step('mobilenetv2', 129, 129, 21, 2, monkeypatch, {})
    off = _step('mobilenetv2', 129, 129, 21, 2, monkeypatch, {'DL3P_BATCHED_WGRAD': '0'})
    monkeypatch.delenv('DL3P_BATCHED_WGRAD')
    assert 'dl3p_irb_fwd' in on[3] and 'dl3p_irb_bwd_data' in on[3]
    assert not any(c.startswith('dl3p_irb_') for c in off[3])
    _same_step(on, off, 3e-2, 0.35)


@pytest.mark.parametrize('model_type,N', [('mobilenetv2', 5), ('xception', 5)])
def test_pinned_schedule_forward_is_taken_at_its_row_threshold_and_the_step_stays(model_type, N, monkeypatch):
    """round 6 (csrc/pw_split3.hip): at 513 x 513 from batch 4 up the decoder's 129 x 129 maps cross 65536 rows and the 256-column
    forwards that the measured table does not know run on the pinned-schedule kernel -- the step is the tiled kernels' step to the
    bound of two summation orders (the kernels agree to 2e-5 per element: tests/test_split_gemm_gpu.py)"""
    import ctypes
    pkg = load_pkg()
    L = load_pkg('ops').lib()
    q = (ctypes.c_int * 6)()
    L.set_option(b'pw_small_min_rows', -1)
    try:
        L.gemm_plan_query(6, N * 129 * 129, 304, 256, q)
        assert q[0] == 3 and q[3] == 4, list(q)
        on = _step(model_type, 513, 513, 21, N, monkeypatch, {}, options={b'pw_small_min_rows': -1})
        off = _step(model_type, 513, 513, 21, N, monkeypatch, {}, options={b'pw_small_min_rows': -1, b'sb3': 0})
    finally:
        L.set_option(b'pw_small_min_rows', 64)
        L.set_option(b'sb3', -1)
    _same_step(on, off, 3e-2, 0.35)
