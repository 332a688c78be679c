"""dl3p_bn_bwd_fused / dl3p_bn_bwd_fused_bf16 (csrc/bn_bwd_fused.hip): the BatchNorm backward of a small tensor -- sums, coefficients
and dz -- in ONE launch with the rows kept in registers across a slab barrier (the gradient of CustomBatchNormalization,
/root/reference deeplabv3p/models/layers.py:63-70).

Checked against a float64 restatement on the device (dz, dgamma, dbeta, the coefficient triple), against the three-launch chain it
replaces (dl3p_bn_bwd_reduce + dl3p_bn_bwd_finalize + dl3p_bn_bwd_apply: same numbers up to the summation order), for run-to-run bit
identity, for the workspace contract (counters back at zero, error word clear, many launches back to back), in place and through
channel-slice views, and inside the model: the step of an executor that traces it against one that does not."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
import test_model_gpu as TM

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    import importlib
    return importlib.import_module(load_pkg().__name__ + '.ops')


def _act_grad64(u, act):
    """TF conventions (0 at the kinks), as csrc/common.h act_grad"""
    one, zero = torch.ones_like(u), torch.zeros_like(u)
    if act == 0:
        return one
    if act == 1:
        return torch.where(u > 0, one, zero)
    if act == 2:
        return torch.where((u > 0) & (u < 6), one, zero)
    t = u + 3
    inside = torch.where((t > 0) & (t < 6), one / 6, zero)
    if act == 4:
        return inside
    return torch.clamp(t, 0, 6) / 6 + u * inside          # hard-swish


def _kink_distance(u, act):
    if act == 0:
        return torch.full_like(u, 1e9)
    if act == 1:
        return u.abs()
    if act == 2:
        return torch.minimum(u.abs(), (u - 6).abs())
    return torch.minimum((u + 3).abs(), (u - 3).abs())


def _state(ops, C, z64, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    bn = ops.BNState(C, DEV, 1e-3)
    bn.gamma.copy_(torch.rand(C, generator=g) + 0.5)
    bn.beta.copy_(torch.randn(C, generator=g) * 0.2)
    mean = z64.mean(0)
    var = z64.var(0, unbiased=False)
    bn.mean.copy_(mean.float())
    bn.invstd.copy_((1.0 / torch.sqrt(var + 1e-3)).float())
    bn.scale.copy_(bn.gamma * bn.invstd)
    bn.shift.copy_(bn.beta - bn.mean * bn.scale)
    return bn


def _reference(bn, g, z, act):
    """float64 on the device from the tensors AS STORED and the fp32 coefficients the kernel reads"""
    g64, z64 = g.double(), z.double()
    u = z64 * bn.scale.double() + bn.shift.double()
    d = g64 * _act_grad64(u, act)
    xh = (z64 - bn.mean.double()) * bn.invstd.double()
    M = z.shape[0]
    s, sx = d.sum(0), (d * xh).sum(0)
    c0 = (bn.gamma * bn.invstd).double()
    dz = c0 * (d - s / M - xh * (sx / M))
    return dz, s, sx, (d.abs().sum(0), (d * xh).abs().sum(0)), _kink_distance(u, act)


FP32_CASES = [(16 * 33 * 33, 256, 1, 0), (4 * 33 * 33, 728, 0, 0), (33 * 33, 2048, 2, 0), (16 * 33 * 33, 48, 2, 8), (5, 4, 2, 0),
              (70001, 16, 3, 0), (300, 12, 1, 4), (2 * 65 * 65, 144, 2, 0), (1, 64, 0, 0), (4097, 20, 4, 0)]
BF16_CASES = [(64 * 128, 960, 3, 0), (128 * 256, 240, 1, 0), (256 * 512, 64, 1, 0), (512 * 1024, 16, 3, 0), (64 * 128, 672, 3, 8),
              (2 * 17 * 19, 72, 2, 0), (100, 20, 2, 0), (64 * 128, 184, 3, 0), (7, 8, 0, 0), (128 * 256, 120, 4, 0)]


def _run_case(ops, M, C, act, pad, bf16, seed=5):
    torch.manual_seed(seed)
    dt = torch.bfloat16 if bf16 else torch.float32
    zbuf = (torch.randn(M, C + pad, device=DEV) * 2 + 0.4).to(dt)
    gbuf = torch.randn(M, C + pad, device=DEV).to(dt)
    z, g = zbuf[:, :C], gbuf[:, :C]
    bn = _state(ops, C, z.double(), seed)
    assert ops.bn_backward_fused_supported(z, bf16), (M, C)
    dz_ref, s, sx, (sa, sxa), kd = _reference(bn, g, z, act)
    out = torch.full((M, C + pad), 7.0, device=DEV).to(dt)
    ops.bn_backward_fused(bn, g, z, act, out=out[:, :C])
    ws = ops.bn_fused_workspace(torch.device(DEV))
    assert ops.lib().bn_bwd_fused_error(ws.data_ptr(), None) == 0
    assert int(ws[:1040].abs().sum()) == 0, 'arrival counters are re-armed by the launch'
    if pad:
        assert bool((out[:, C:] == 7.0).all()), 'columns beyond C are not touched'
    # sums: float32 per-thread accumulation -> relative to the sum of |terms|
    assert float(((bn.dbeta.double() - s).abs() / (sa + 1e-30)).max()) < 2e-6 * max(1.0, np.sqrt(M) / 64), 'dbeta'
    assert float(((bn.dgamma.double() - sx).abs() / (sxa + 1e-30)).max()) < 2e-6 * max(1.0, np.sqrt(M) / 64), 'dgamma'
    coef = bn.coef.view(3, C).double()
    assert float((coef[0] - (bn.gamma * bn.invstd).double()).abs().max()) == 0.0
    assert float(((coef[1] - s / M).abs() * M / (sa + 1e-30)).max()) < 4e-6 * max(1.0, np.sqrt(M) / 64)
    assert float(((coef[2] - sx / M).abs() * M / (sxa + 1e-30)).max()) < 4e-6 * max(1.0, np.sqrt(M) / 64)
    # dz: elements whose pre-activation sits on a kink of the activation within float32 rounding may take either branch
    clear = kd > 1e-4
    err = (out[:, :C].double() - dz_ref).abs()
    scale = dz_ref.abs().max()
    tol = (2.0 ** -8 if bf16 else 2e-6) * (dz_ref.abs() + 1e-3 * scale) + (1e-6 * scale)
    bad = (err > tol) & clear
    assert not bool(bad.any()), (int(bad.sum()), float((err * clear).max()), float(scale))
    return bn, g, z, out


@pytest.mark.parametrize('M,C,act,pad', FP32_CASES)
def test_fused_bn_backward_fp32_vs_float64(ops, M, C, act, pad):
    _run_case(ops, M, C, act, pad, False)


@pytest.mark.parametrize('M,C,act,pad', BF16_CASES)
def test_fused_bn_backward_bf16_vs_float64(ops, M, C, act, pad):
    _run_case(ops, M, C, act, pad, True)


@pytest.mark.parametrize('bf16', [False, True])
def test_fused_matches_the_three_launch_chain_and_is_bit_reproducible(ops, bf16):
    M, C, act = (64 * 128, 480, 3) if bf16 else (16 * 33 * 33, 256, 2)
    bn, g, z, out = _run_case(ops, M, C, act, 0, bf16, seed=9)
    first = (out.clone(), bn.dgamma.clone(), bn.dbeta.clone(), bn.coef.clone())
    for _ in range(25):                       # back to back on one stream: the counters are re-armed every time
        ops.bn_backward_fused(bn, g, z, act, out=out)
    assert all(bool((a == b).all()) for a, b in zip(first, (out, bn.dgamma, bn.dbeta, bn.coef))), 'run-to-run bit identity'
    inplace = g.clone()
    ops.bn_backward_fused(bn, inplace, z, act)        # dz over g
    assert bool((inplace == first[0]).all()), 'in place'
    part = ops.new_partials(C, DEV)
    sep = g.clone()
    if bf16:
        ops.bn_backward_bf16(bn, sep, z, act, part)
    else:
        ops.bn_backward(bn, sep, z, act, part)
    # the chain sums the same float32 terms in another order
    a, b = first[0].double(), sep.double()
    assert float((a - b).abs().max()) <= (2.0 ** -7 if bf16 else 2e-6) * float(b.abs().max())
    assert float(((bn.dgamma - first[1]).abs() / (first[1].abs() + 1e-2 * first[1].abs().max())).max()) < 2e-5
    assert float(((bn.dbeta - first[2]).abs() / (first[2].abs() + 1e-2 * first[2].abs().max())).max()) < 2e-5


def test_supported_rule_and_errors(ops):
    L = ops.lib()
    assert L.bn_bwd_fused_supported(64 * 128, 960, 2, 960, 960, 960) == 1
    assert L.bn_bwd_fused_supported(512 * 1024, 16, 2, 16, 16, 16) == 1
    assert L.bn_bwd_fused_supported(512 * 1024, 64, 2, 64, 64, 64) == 0, '33 M elements do not fit one resident grid'
    assert L.bn_bwd_fused_supported(16 * 33 * 33, 256, 4, 256, 256, 256) == 1
    assert L.bn_bwd_fused_supported(16 * 129 * 129, 256, 4, 256, 256, 256) == 0
    assert L.bn_bwd_fused_supported(100, 6, 4, 8, 8, 8) == 0, 'C % 4'
    z = torch.randn(1000, 32, device=DEV)
    bn = _state(ops, 32, z.double(), 1)
    ws = torch.zeros(64, dtype=torch.int32, device=DEV)
    with pytest.raises(Exception, match='workspace'):
        L.bn_bwd_fused(z.data_ptr(), 32, z.data_ptr(), 32, bn.scale.data_ptr(), bn.shift.data_ptr(), 1, bn.mean.data_ptr(),
                       bn.invstd.data_ptr(), bn.gamma.data_ptr(), None, None, None, z.data_ptr(), 32, 1000, 32, ws.data_ptr(), 256, None)
    big = torch.zeros(16 * 129 * 129, 256, device=DEV)
    with pytest.raises(Exception, match='resident grid'):
        ops.bn_backward_fused(_state(ops, 256, big[:64].double(), 1), big, big, 1)


def _step(model_type, H, W, C, N, monkeypatch, fused, bf16=False):
    import importlib
    pkg = load_pkg()
    mp = importlib.import_module(pkg.__name__ + '.mixed_precision')
    monkeypatch.setenv('DL3P_BN_BWD_FUSED', '1' if fused else '0')
    if bf16:
        mp.set_global_policy('mixed_bfloat16')
    try:
        torch.manual_seed(0)
        x, y = TM._data(N, H, W, C, seed=29)
        m = pkg.get_deeplabv3p_model(model_type, C, (H, W), 16, training=True, seed=3)
        m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
        m.use_graphs = False
        loss = m.train_on_batch(x, y)
        st = m._store
        ex = m._executor(N, True)
        calls = [ep for plan in (ex.fwd, ex.bwd) for (ep, _) in plan.labels]
        g = {p.name: np.array(st.get(p, st.G), dtype=np.float64) for p in m.graph.all_params() if p.trainable}
        if fused and getattr(ex, 'bn_fused_ws', None) is not None:
            assert ex.L.bn_bwd_fused_error(ex.bn_fused_ws.data_ptr(), None) == 0
        return loss, g, calls
    finally:
        if bf16:
            mp.set_global_policy('float32')


@pytest.mark.parametrize('model_type,H,W,N,bf16', [('mobilenetv2', 129, 129, 2, False), ('xception', 129, 129, 2, False),
                                                    ('mobilenetv3large', 128, 256, 1, True)])
def test_the_executor_traces_the_one_launch_form_and_the_step_stays(monkeypatch, model_type, H, W, N, bf16):
    on = _step(model_type, H, W, 21, N, monkeypatch, True, bf16)
    off = _step(model_type, H, W, 21, N, monkeypatch, False, bf16)
    name = 'dl3p_bn_bwd_fused_bf16' if bf16 else 'dl3p_bn_bwd_fused'
    n_on = sum(1 for c in on[2] if c == name)
    assert n_on >= 5 and not any(c == name for c in off[2]), (n_on, sorted(set(on[2])))
    assert len(on[2]) <= len(off[2]) - 2 * n_on, 'each fused BatchNorm replaces its reduce / finalize / apply launches'
    (l1, g1, _), (l0, g0, _) = on, off
    tol_l, tol_g = (2e-3, 3e-2) if bf16 else (2e-6, 2e-4)
    assert abs(l1 - l0) <= tol_l * abs(l0), (l1, l0)
    num = sum(float(((v - g1[n]) ** 2).sum()) for n, v in g0.items())
    den = sum(float((v ** 2).sum()) for v in g0.values())
    assert np.sqrt(num / den) < tol_g, np.sqrt(num / den)
