"""Decoder_block without the resized tensor in HBM (deeplabv3p/models/layers.py:207-215; dl3p_dw_upsampled_input, executor._find_up):
the 3x3 depthwise conv forms the first channels of its input from the low-resolution map while it loads -- the resize kernel's
arithmetic expression for expression, so everything downstream equals the unfused launches BIT FOR BIT."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
import test_model_gpu as TM

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('case', [(2, 9, 9, 33, 33, 32, 16), (16, 33, 33, 129, 129, 256, 48), (1, 5, 7, 19, 26, 8, 4), (3, 17, 17, 65, 65, 256, 48),
                                  (2, 33, 33, 129, 129, 304, 0)])
def test_upsampled_input_equals_resize_then_conv(ops, case):
    N, h, w, H, W, C1, C2 = case
    C = C1 + C2
    g = torch.Generator(device=DEV); g.manual_seed(H * 7 + C)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    low = rnd(N, h, w, C1).abs()
    buf = rnd(N, H, W, C)                     # channels [C1, C) = the skip features; [0, C1) filled by the resize (unfused) or never read
    wk = rnd(3, 3, C) * 0.3
    sc = torch.ones(C, device=DEV); sh = torch.zeros(C, device=DEV)
    sc[C1:] = torch.rand(C2, device=DEV, generator=g) + 0.5
    sh[C1:] = rnd(C2) * 0.3
    L = ops.lib()
    geo = (N, H, W, C, C1, 3, 1, 1, 1, 1, H, W)
    assert L.dw_upsampled_input_supported(0, *geo) and L.dw_upsampled_input_supported(1, *geo)
    ref_in = buf.clone()
    ops.resize_bilinear_fwd(low, H, W, out=ref_in[..., :C1])
    part0, part1 = ops.new_partials(C, DEV), ops.new_partials(C, DEV)
    y0, r0 = ops.dwconv2d_fwd(ref_in, wk, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU, partials=part0)
    poisoned = buf.clone()
    poisoned[..., :C1] = float('nan')          # the fused launch must not read these
    y1, r1 = ops.dwconv2d_fwd(poisoned, wk, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU, partials=part1, upsampled=low)
    assert r0 == r1 and torch.equal(y0, y1)
    assert torch.equal(part0[:r0 * 2 * C], part1[:r1 * 2 * C])
    # ... and against float64
    a64 = torch.relu(ref_in.double() * sc.double() + sh.double())
    y64 = torch.nn.functional.conv2d(a64.permute(0, 3, 1, 2), wk.double().permute(2, 0, 1).unsqueeze(1), padding=1, groups=C).permute(0, 2, 3, 1)
    assert float((y1.double() - y64).abs().max()) < 2e-5 * float(y64.abs().max())
    # weight gradient, plain and with the BatchNorm-backward apply folded in
    dy = rnd(N, H, W, C)
    gw0 = ops.dwconv2d_bwd_weight(ref_in, dy, 3, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU)
    gw1 = ops.dwconv2d_bwd_weight(poisoned, dy, 3, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU, upsampled=low)
    assert torch.equal(gw0, gw1)
    bn = ops.BNState(C, DEV)
    bn.gamma.copy_(torch.rand(C, device=DEV, generator=g) + 0.5); bn.beta.copy_(rnd(C) * 0.2)
    z = y0
    bn.mean.copy_(z.mean((0, 1, 2))); bn.invstd.copy_(1.0 / torch.sqrt(z.var((0, 1, 2), unbiased=False) + bn.eps))
    bn.scale.copy_(bn.gamma * bn.invstd); bn.shift.copy_(bn.beta - bn.mean * bn.scale)
    bn.coef.copy_(torch.cat([bn.scale, rnd(C) * 0.01, rnd(C) * 0.01]))
    g0, dz0 = ops.dwconv2d_bwd_weight_bn(ref_in, dy, z, bn, ops.ACT_RELU, 3, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU)
    g1, dz1 = ops.dwconv2d_bwd_weight_bn(poisoned, dy, z, bn, ops.ACT_RELU, 3, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU, upsampled=low)
    assert torch.equal(g0, g1) and torch.equal(dz0, dz1)
    # the description is consumed by ONE call
    y2 = ops.dwconv2d_fwd(ref_in, wk, in_scale=sc, in_shift=sh, in_act=ops.ACT_RELU)
    assert torch.equal(y2, y0)


def test_unserved_geometries_are_refused(ops):
    L = ops.lib()
    assert not L.dw_upsampled_input_supported(0, 2, 33, 33, 64, 32, 3, 2, 1, 0, 0, 17, 17)       # stride 2
    assert not L.dw_upsampled_input_supported(0, 2, 33, 33, 64, 32, 3, 1, 6, 6, 6, 33, 33)       # atrous
    assert not L.dw_upsampled_input_supported(0, 2, 33, 33, 64, 32, 5, 1, 1, 2, 2, 33, 33)       # 5 x 5
    low = torch.zeros(2, 9, 9, 32, device=DEV)
    x = torch.zeros(2, 33, 33, 64, device=DEV)
    w = torch.zeros(3, 3, 64, device=DEV)
    with pytest.raises(ops.Dl3pError):
        ops.dwconv2d_fwd(x, w, upsampled=low)                         # no BatchNorm + activation prologue
    assert torch.equal(ops.dwconv2d_fwd(x, w), torch.zeros_like(x))   # (the failed call consumed the description)


@pytest.mark.parametrize('model_type,H,W,N', [('mobilenetv2', 129, 129, 2), ('xception', 65, 97, 2), ('mobilenetv2', 513, 513, 2)])
def test_the_step_with_the_resize_folded_is_the_unfolded_step_bit_for_bit(model_type, H, W, N, monkeypatch):
    C = 21
    x, y = TM._data(N, H, W, C, seed=31)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(0)
        m, _ = TM._pair(model_type, H, W, C)
        losses = [m.train_on_batch(x, y) for _ in range(2)]
        ex = m._executor(N, True)
        calls = [ep for plan in (ex.fwd, ex.bwd) for (ep, _) in plan.labels]
        st = m._store
        g = {p.name: np.array(st.get(p, st.G)) for p in m.graph.all_params() if p.trainable}
        w = {k: np.array(v) for k, v in m.get_weights_by_name().items()}
        p = m.predict(x)
        for k in env:
            monkeypatch.delenv(k)
        del m
        torch.cuda.empty_cache()
        return losses, g, w, calls, p
    l1, g1, w1, c1, p1 = run({'DL3P_FOLD_RESIZE': '1'})          # (opt-in: measured slower than the unfused pair, executor._find_up)
    l0, g0, w0, c0, p0 = run({})
    assert c1.count('dl3p_dw_upsampled_input') == 2 and 'dl3p_dw_upsampled_input' not in c0
    assert c0.count('dl3p_resize_bilinear_fwd') == c1.count('dl3p_resize_bilinear_fwd') + 1
    assert l1 == l0, (l1, l0)
    assert all(np.array_equal(g0[k], g1[k]) for k in g0)
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)
    assert np.array_equal(p0, p1)
