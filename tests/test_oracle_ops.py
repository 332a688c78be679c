"""The CPU oracle (oracle/np_ops.py) triangulated against an independent implementation (torch CPU
ops).  TensorFlow is not installable here, so two independent implementations agreeing on the TF
semantics spelled out in SURVEY.md section 8c is the available substitute for reference outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ops as O

RNG = np.random.default_rng(0)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def nchw(a):
    return t(a).permute(0, 3, 1, 2)


def nhwc(tt):
    return tt.permute(0, 2, 3, 1).numpy()


def torch_pad(x, pads):
    pt, pb, pl, pr = pads
    return F.pad(x, (pl, pr, pt, pb))


@pytest.mark.parametrize('H,W,k,s,r,padding', [
    (33, 33, 3, 1, 18, 'same'), (33, 33, 3, 1, 6, 'same'), (17, 19, 3, 1, 2, 'same'), (16, 20, 3, 2, 1, 'same'),
    (33, 33, 3, 2, 1, 'same'), (16, 24, 5, 2, 1, 'same'), (16, 24, 5, 1, 2, 'same'), (33, 33, 3, 2, 1, (1, 1, 1, 1)),
    (12, 12, 3, 2, 2, (2, 2, 2, 2))])
def test_depthwise_matches_torch(H, W, k, s, r, padding):
    C = 6
    x = RNG.standard_normal((2, H, W, C))
    w = RNG.standard_normal((k, k, C))
    Ho, Wo, pads = O.resolve_padding(H, W, k, s, r, padding)
    y = O.dwconv2d_fwd(x, w, s, r, padding)
    xt = nchw(x).requires_grad_(True)
    wt = t(w).permute(2, 0, 1).unsqueeze(1).contiguous().requires_grad_(True)   # (C,1,k,k)
    yt = F.conv2d(torch_pad(xt, pads), wt, stride=s, dilation=r, groups=C)
    assert yt.shape[2:] == (Ho, Wo)
    np.testing.assert_allclose(y, nhwc(yt.detach()), atol=1e-10)
    gy = RNG.standard_normal(y.shape)
    yt.backward(nchw(gy))
    gx, gw = O.dwconv2d_bwd(x, w, gy, s, r, padding)
    np.testing.assert_allclose(gx, nhwc(xt.grad), atol=1e-10)
    np.testing.assert_allclose(gw, wt.grad[:, 0].permute(1, 2, 0).numpy(), atol=1e-9)


def test_same_padding_is_asymmetric_at_even_sizes():
    # TF SAME: the odd unit of padding goes to the END (bottom/right) -- config C5 (1024x2048)
    assert O.same_pad_1d(16, 3, 2, 1) == (8, 0, 1)
    assert O.same_pad_1d(16, 5, 2, 1) == (8, 1, 2)
    assert O.same_pad_1d(513, 3, 2, 1) == (257, 1, 1)
    assert O.same_pad_1d(33, 3, 1, 18) == (33, 18, 18)


@pytest.mark.parametrize('k,s,padding,cin,cout', [(3, 2, 'same', 3, 8), (3, 1, 'same', 4, 6), (1, 1, 'same', 8, 5),
                                                  (1, 2, (0, 0, 0, 0), 8, 4), (3, 2, (0, 1, 0, 1), 3, 4)])
def test_dense_conv_matches_torch(k, s, padding, cin, cout):
    H, W = 17, 16
    x = RNG.standard_normal((2, H, W, cin))
    w = RNG.standard_normal((k, k, cin, cout))
    b = RNG.standard_normal(cout)
    Ho, Wo, pads = O.resolve_padding(H, W, k, s, 1, padding)
    y = O.conv2d_fwd(x, w, s, 1, padding, b)
    xt = nchw(x).requires_grad_(True)
    wt = t(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    bt = t(b).requires_grad_(True)
    yt = F.conv2d(torch_pad(xt, pads), wt, bt, stride=s)
    np.testing.assert_allclose(y, nhwc(yt.detach()), atol=1e-10)
    gy = RNG.standard_normal(y.shape)
    yt.backward(nchw(gy))
    gx, gw, gb = O.conv2d_bwd(x, w, gy, s, 1, padding)
    np.testing.assert_allclose(gx, nhwc(xt.grad), atol=1e-10)
    np.testing.assert_allclose(gw, wt.grad.permute(2, 3, 1, 0).numpy(), atol=1e-9)
    np.testing.assert_allclose(gb, bt.grad.numpy(), atol=1e-9)


def test_batchnorm_matches_torch():
    x = RNG.standard_normal((3, 5, 7, 4)) * 2 + 1
    g, b = RNG.uniform(0.5, 1.5, 4), RNG.standard_normal(4)
    eps, mom = 1e-3, 0.99
    y, cache, (bm, bv) = O.bn_train_fwd(x, g, b, eps)
    xt = nchw(x).requires_grad_(True)
    gt, bt = t(g).requires_grad_(True), t(b).requires_grad_(True)
    rm, rv = torch.zeros(4).double(), torch.ones(4).double()
    # torch momentum is (1 - keras momentum); torch feeds the UNBIASED variance into the moving average, Keras'
    # SyncBatchNormalization (the reference's CustomBatchNormalization) the biased one
    yt = F.batch_norm(xt, rm, rv, gt, bt, training=True, momentum=1 - mom, eps=eps)
    np.testing.assert_allclose(y, nhwc(yt.detach()), atol=1e-10)
    np.testing.assert_allclose(O.bn_moving_update(np.zeros(4), bm, mom), rm.numpy(), atol=1e-12)
    m = x.size // 4
    np.testing.assert_allclose(O.bn_moving_update(np.ones(4), bv * m / (m - 1), mom), rv.numpy(), atol=1e-12)
    # the two rules of np_ops.bn_moving_variance_of: 'unbiased' IS what a fused batch-norm kernel feeds its running
    # variance (torch's does the same as TF's FusedBatchNormV3), 'biased' the plain E[x^2] - E[x]^2 of Keras' non-fused path
    np.testing.assert_allclose(O.bn_moving_update(np.ones(4), O.bn_moving_variance_of(bv, m, 'unbiased'), mom), rv.numpy(), atol=1e-12)
    np.testing.assert_allclose(O.bn_moving_variance_of(bv, m, 'biased'), x.reshape(-1, 4).var(0), atol=1e-12)
    np.testing.assert_allclose(O.bn_moving_variance_of(bv, m, 'unbiased'), x.reshape(-1, 4).var(0, ddof=1), atol=1e-12)
    gy = RNG.standard_normal(y.shape)
    yt.backward(nchw(gy))
    gx, gg, gb = O.bn_train_bwd(gy, cache)
    np.testing.assert_allclose(gx, nhwc(xt.grad), atol=1e-10)
    np.testing.assert_allclose(gg, gt.grad.numpy(), atol=1e-9)
    np.testing.assert_allclose(gb, bt.grad.numpy(), atol=1e-9)
    yi = O.bn_infer_fwd(x, g, b, bm, bv, eps)
    yti = F.batch_norm(nchw(x), t(bm), t(bv), t(g), t(b), training=False, eps=eps)
    np.testing.assert_allclose(yi, nhwc(yti), atol=1e-10)


@pytest.mark.parametrize('h,w,H,W', [(33, 33, 129, 129), (1, 1, 33, 33), (129, 129, 513, 513), (9, 13, 33, 50),
                                     (16, 32, 64, 128), (10, 10, 7, 5)])
def test_bilinear_matches_torch_half_pixel(h, w, H, W):
    """tf.image.resize(bilinear) in TF2 (half_pixel_centers, no antialias) == F.interpolate(align_corners=False)
    for up-sampling; for down-sampling torch uses the same kernel (antialias=False)"""
    x = RNG.standard_normal((2, h, w, 3))
    y = O.resize_bilinear_fwd(x, H, W)
    xt = nchw(x).requires_grad_(True)
    yt = F.interpolate(xt, size=(H, W), mode='bilinear', align_corners=False)
    np.testing.assert_allclose(y, nhwc(yt.detach()), atol=2e-4)   # coordinates are float32 like TF's (torch: float64)
    gy = RNG.standard_normal(y.shape)
    yt.backward(nchw(gy))
    np.testing.assert_allclose(O.resize_bilinear_bwd(gy, h, w), nhwc(xt.grad), atol=2e-3)


def test_softmax_ce_matches_torch_and_keras_reduction():
    C = 21
    z = RNG.standard_normal((2, 5, 5, C)) * 3
    lab = RNG.integers(0, C, (2, 5, 5)).astype(np.float64)
    lab[0, 0, :3] = 255
    loss, p, g = O.sparse_ce_fwd_bwd(z, lab, 255)
    zt = t(z).requires_grad_(True)
    lt = torch.from_numpy(lab).long()
    # Keras: mean over ALL pixels (ignored ones stay in the denominator)
    lsum = F.cross_entropy(zt.reshape(-1, C), lt.reshape(-1), ignore_index=255, reduction='sum') / lab.size
    np.testing.assert_allclose(loss, lsum.item(), rtol=1e-10)
    lsum.backward()
    np.testing.assert_allclose(g, zt.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(p.sum(-1), 1.0, atol=1e-12)
    # ignore_index=0 is falsy in the reference (loss.py:139): nothing is masked, 255 rows are all-zero one-hots
    loss0, _, g0 = O.sparse_ce_fwd_bwd(z, lab, 0)
    np.testing.assert_allclose(loss0, loss, rtol=1e-12)


def test_activations_and_grads():
    x = RNG.standard_normal(1000) * 4
    xt = t(x).requires_grad_(True)
    for act, f in [(O.ACT_RELU, F.relu), (O.ACT_RELU6, F.relu6), (O.ACT_HSWISH, F.hardswish), (O.ACT_HSIGMOID, F.hardsigmoid)]:
        y = f(xt)
        np.testing.assert_allclose(O.act_fwd(x, act), y.detach().numpy(), atol=1e-12)
        g, = torch.autograd.grad(y.sum(), xt)
        np.testing.assert_allclose(O.act_bwd(x, np.ones_like(x), act), g.numpy(), atol=1e-12)


def test_sgd_matches_keras_rule():
    w, v, g = RNG.standard_normal(10), RNG.standard_normal(10), RNG.standard_normal(10)
    w2, v2 = O.sgd_momentum_step(w, v, g, 0.01, 0.9, 2e-5)
    gt = g + 2 * 2e-5 * w
    np.testing.assert_allclose(v2, 0.9 * v - 0.01 * gt)
    np.testing.assert_allclose(w2, w + v2)


def test_oracle_model_gradient_check():
    """finite-difference check of the tape through a whole (tiny-input) MobileNetV2-lite graph"""
    from oracle.np_net import OracleModel
    m = OracleModel('mobilenetv2_lite', 5, (33, 33), 16, dtype=np.float64, seed=1)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (2, 33, 33, 3))
    y = rng.integers(0, 5, (2, 33 * 33, 1)).astype(np.float64)
    y[0, :40] = 255
    total, ce, _ = m.loss_and_grads(x, y)
    # (parameters deep in the backbone sit behind dozens of ReLU6 kinks on 3x3 feature maps: their finite
    # differences only converge at eps ~1e-8; every op's backward is checked exactly against torch above)
    for name in ['conv_upsample/kernel', 'conv_upsample/bias', 'concat_projection/kernel',
                 'concat_projection_BN/gamma', 'aspp0/kernel']:
        g = m.net.grads[name]
        d = rng.standard_normal(g.shape)
        eps = 1e-6   # small enough not to cross ReLU kinks
        w0 = m.net.params[name].copy()
        m.net.params[name] = w0 + eps * d
        _, cp, _ = m.loss_and_grads(x, y)
        m.net.params[name] = w0 - eps * d
        _, cm, _ = m.loss_and_grads(x, y)
        m.net.params[name] = w0
        fd = (cp - cm) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) < 1e-3 * max(1.0, abs(an)), (name, fd, an)


@pytest.mark.parametrize('mt', ['mobilenetv2_lite', 'mobilenetv3large'])
def test_whole_model_matches_torch_autograd(mt):
    """The oracle's whole train step (forward, hand-written reverse-mode tape, CE with ignore 255) against an independent
    implementation of every primitive AND of differentiation: oracle/torch_net.py runs the same builders on
    torch.nn.functional ops + torch autograd (fp64).  Activations agree to 1e-11; the final bilinear pred_resize carries
    torch's float32 interpolation weights under autograd (6e-7), which bounds loss / gradient agreement at 1e-5."""
    from oracle.np_net import OracleModel
    from oracle.torch_net import TorchModel
    H = W = 33
    N, C = 2, 5
    o = OracleModel(mt, C, (H, W), 16, dtype=np.float64, seed=0)
    t = TorchModel(mt, C, (H, W), 16, dtype=np.float64, seed=0)
    assert list(o.net.order) == list(t.net.order)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (N, H, W, 3))
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float64)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    mask = (rng.uniform(size=(N, 3, 3, 256)) >= 0.5).astype(np.float64)
    lo, po = o.predict(x)
    lt, pt = t.predict(x)
    np.testing.assert_allclose(lo, lt, atol=1e-10, rtol=0)   # inference: every op
    _, co, _ = o.loss_and_grads(x, y, {'aspp_dropout': mask})
    _, ct, _ = t.loss_and_grads(x, y, {'aspp_dropout': mask})
    # (behind a decoder_resize / pred_resize the torch side carries its float32 bilinear weights)
    for k, tol in (('backbone_out', 1e-10), ('head_in', 5e-6), ('conv_upsample', 5e-6)):
        np.testing.assert_allclose(o.net.taps[k].v, t.net.taps[k].v.detach().numpy(), atol=tol, rtol=0, err_msg=k)
    assert abs(co - ct) < 1e-7
    for k, g in o.net.grads.items():
        if np.abs(g).max() > 1e-7:                            # (a beta in front of conv + BN has an exactly-zero gradient)
            assert np.abs(g - t.net.grads[k]).max() < 1e-5 * np.abs(g).max(), k


def test_moving_variance_rule_reaches_the_model():
    """OracleModel(bn_moving_variance=...): one train step moves every moving_variance by the rule chosen, the
    two differ by exactly count / (count - 1) in the batch term, and nothing else (loss, gradients, moving means) differs"""
    from oracle.np_net import OracleModel
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (2, 33, 33, 3))
    y = rng.integers(0, 5, (2, 33 * 33, 1)).astype(np.float64)
    res = {}
    for rule in ('biased', 'unbiased'):
        o = OracleModel('mobilenetv2_lite', 5, (33, 33), 16, dtype=np.float64, seed=0, bn_moving_variance=rule)
        mv0 = {k: v.copy() for k, v in o.net.params.items() if k.endswith('moving_variance')}
        out = o.loss_and_grads(x, y, {})
        o.sgd_step(0.01, 0.9)
        res[rule] = (out[0], {k: v.copy() for k, v in o.net.grads.items()}, dict(o.net.params), mv0)
    (la, ga, pa, mv0), (lb, gb, pb, _) = res['biased'], res['unbiased']
    assert la == lb and all(np.array_equal(ga[k], gb[k]) for k in ga)
    checked = 0
    for k in pa:
        if k.endswith('moving_variance'):
            mom = 0.999 if k.startswith(('Conv', 'expanded_conv', 'bn_Conv1')) else 0.99
            batch_a = (pa[k] - mom * mv0[k]) / (1 - mom)
            batch_b = (pb[k] - mom * mv0[k]) / (1 - mom)
            # image_pooling_BN normalises N = 2 samples per channel: count / (count - 1) = 2
            count = 2 if k.startswith('image_pooling') else None
            ratio = batch_b[batch_a > 1e-12] / batch_a[batch_a > 1e-12]
            if count:
                np.testing.assert_allclose(ratio, 2.0, rtol=1e-6)
                checked += 1
            else:
                assert np.all(ratio > 1.0) and np.all(ratio < 1.13), (k, ratio.min(), ratio.max())
        else:
            assert np.array_equal(pa[k], pb[k]), k
    assert checked == 1
