"""a cross-check for the graphs no third-party port in this image covers -- the Xception body, the MobileNetV3
bodies with squeeze-excite, the SepConv ASPP and the decoder -- against tests/indep_torch_graphs.py, a torch.nn.Module tree written
from the reference's model files without any of the oracle's builders, padding helpers, resize or loss (see its header), sharing
only the weights.  Inference mode: every convolution's output, layer by layer, at output strides 16 and 8 and odd / even sizes.
Training mode: loss and every parameter gradient (torch autograd against the oracle's hand-written tape), BatchNorm on batch
statistics, the dropout mask supplied.  It does not pin the oracle (nothing but TensorFlow can) -- it widens the
evidence: a restatement error would have to be made identically in two differently-shaped programs."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

import indep_torch_graphs as G  # noqa: E402
from indep_torch_graphs import DeepLabV3Plus, keras_sparse_ce, tf_same_pad, resize_matrix  # noqa: E402


# The oracle's bilinear resize follows TensorFlow's kernel, which computes the sample coordinate and the interpolation weight in
# float32 (oracle/np_ops.py bilinear_coeffs); the independent graph uses the real-valued weights.  From decoder_resize on (17 -> 33
# and the like: coordinates that are not dyadic) the two differ by the float32 rounding of the sample coordinate, 2^-24 of the input
# extent in the weight.
BEHIND_DECODER_RESIZE = {'decoder_conv0_depthwise', 'decoder_conv0_pointwise', 'decoder_conv1_depthwise', 'decoder_conv1_pointwise',
                         'conv_upsample'}
AFTER_RESIZE = 5e-6


def _randomise_bn(params, rng):
    for k, v in params.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/moving_mean') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.2
        elif k.endswith('/moving_variance'):
            v[...] = rng.uniform(0.5, 2.0, v.shape)


@pytest.mark.parametrize('mt,OS,hw', [('xception', 16, (65, 65)), ('xception', 8, (49, 65)), ('xception', 16, (64, 48)),
                                      ('mobilenetv3large', 16, (65, 65)), ('mobilenetv3large', 8, (64, 96)),
                                      ('mobilenetv3small', 16, (65, 49)), ('mobilenetv3small', 8, (64, 64)),
                                      ('resnet50', 16, (65, 65)), ('resnet50', 8, (64, 80)),
                                      # the BASELINE configs[1] / configs[0] models (transformers covers their body and the lite head)
                                      ('mobilenetv2', 16, (65, 65)), ('mobilenetv2', 8, (48, 64)), ('mobilenetv2_lite', 16, (65, 65)),
                                      ('mobilenetv3large_lite', 16, (64, 64))])
def test_every_convolution_output_equals_the_independent_graph(mt, OS, hw):
    from oracle.np_net import OracleModel
    C = 19 if OS == 8 else 21
    o = OracleModel(mt, C, hw, OS, dtype=np.float64, seed=3)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (2,) + hw + (3,))
    _randomise_bn(o.net.params, rng)
    o.net.record = {}
    logits, probs = o.predict(x)
    rec = o.net.record
    t = DeepLabV3Plus(mt, C, hw, OS).double().eval()
    t.load_keras(o.net.params)
    got, hooks = t.record_convs()
    with torch.no_grad():
        lt = t(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy()))
    for h in hooks:
        h.remove()
    assert set(got) == set(rec), sorted(set(got) ^ set(rec))[:6]
    # same layers in the same ORDER (the order of execution is the order the names were recorded in)
    assert list(got) == list(rec)
    for name, ref in got.items():
        mine = rec[name]
        assert mine.shape == ref.shape, (name, mine.shape, ref.shape)
        err = float(np.abs(mine - ref).max()) / max(1e-30, float(np.abs(ref).max()))
        assert err < (AFTER_RESIZE if name in BEHIND_DECODER_RESIZE else 1e-10), (name, err)
    ref = np.transpose(lt.numpy(), (0, 2, 3, 1))
    assert logits.shape == ref.shape == (2,) + hw + (C,)
    assert float(np.abs(logits - ref).max()) < AFTER_RESIZE * max(1.0, float(np.abs(ref).max()))
    pr = torch.softmax(lt, dim=1).permute(0, 2, 3, 1).numpy()
    assert float(np.abs(probs - pr).max()) < AFTER_RESIZE
    # the output stride is the one asked for
    last = rec['exit_flow_block2_separable_conv3_pointwise' if mt == 'xception' else 'res5c_branch2c' if mt == 'resnet50' else
               'expanded_conv_16_project' if mt.startswith('mobilenetv2') else 'expanded_conv_%d/project' % (14 if 'large' in mt else 10)]
    assert last.shape[1:3] == (-(-hw[0] // OS), -(-hw[1] // OS))


@pytest.mark.parametrize('mt,OS', [('xception', 16), ('mobilenetv3large', 8), ('mobilenetv3small', 16), ('resnet50', 16), ('mobilenetv2', 16), ('mobilenetv2_lite', 16)])
def test_train_step_loss_and_gradients_equal_torch_autograd_on_the_independent_graph(mt, OS, monkeypatch):
    from oracle.np_net import OracleModel
    monkeypatch.setattr(G, 'FLOAT32_COORDS', True)     # (see indep_torch_graphs.py: the resize weights rounded as TF's kernel rounds them)
    H = W = 33
    N, C = 2, 5
    o = OracleModel(mt, C, (H, W), OS, dtype=np.float64, seed=0)
    rng = np.random.default_rng(1)
    _randomise_bn(o.net.params, rng)
    x = rng.uniform(-1, 1, (N, H, W, 3))
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float64)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    fh, fw = -(-H // OS), -(-W // OS)
    mask = (rng.uniform(size=(N, fh, fw, 256)) >= 0.5).astype(np.float64)
    t = DeepLabV3Plus(mt, C, (H, W), OS).double().train()
    t.load_keras(o.net.params)
    mv0 = {k: v.copy() for k, v in o.net.params.items() if k.endswith('moving_mean') or k.endswith('moving_variance')}
    counts = {}
    for m in t.modules():
        if hasattr(m, 'bn'):
            m.bn.register_forward_hook(lambda mod, a, out, name=m.kname: counts.__setitem__(name, a[0].numel() // a[0].shape[1]))
    _, ce_o, logits_o = o.loss_and_grads(x, y, {'aspp_dropout': mask})
    lt = t(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy()), torch.from_numpy(np.transpose(mask, (0, 3, 1, 2)).copy()))
    np.testing.assert_allclose(logits_o, lt.detach().permute(0, 2, 3, 1).numpy(), atol=1e-9, rtol=0)
    ce_t = keras_sparse_ce(lt, torch.from_numpy(y.reshape(N, H, W)))
    assert abs(ce_o - float(ce_t.detach())) < 1e-11
    ce_t.backward()
    gt = t.keras_grads()
    grads = {k: g for k, g in o.net.grads.items()}
    assert set(gt) == set(grads), sorted(set(gt) ^ set(grads))[:6]
    checked = 0
    for k, g in grads.items():
        scale = float(np.abs(g).max())
        if scale > 1e-9:                       # (a bias / beta in front of a batch-statistics BatchNorm has an exactly-zero gradient)
            assert float(np.abs(g - gt[k]).max()) < 1e-7 * scale, k
            checked += 1
        else:
            assert float(np.abs(gt[k]).max()) < 1e-9, k
    assert checked > 0.75 * len(grads)          # (ResNet50: 53 conv biases in front of a BatchNorm)
    # the moving statistics the step leaves behind: Keras' m * old + (1 - m) * batch with the BIASED batch variance (the oracle's
    # default rule), against torch's own running statistics (which take the UNBIASED one: undo count / (count - 1))
    upd = o.net.moving_updates
    bns = {m.kname: m.bn for m in t.modules() if hasattr(m, 'bn')}
    assert set(upd) == {n + s for n in bns for s in ('/moving_mean', '/moving_variance')}
    for name, bn in bns.items():
        np.testing.assert_allclose(upd[name + '/moving_mean'], bn.running_mean.numpy(), atol=1e-10, rtol=0, err_msg=name)
        mom, cnt = bn.momentum, counts[name]
        old = mv0[name + '/moving_variance']
        batch_unbiased = (bn.running_var.numpy() - (1 - mom) * old) / mom
        np.testing.assert_allclose(upd[name + '/moving_variance'], (1 - mom) * old + mom * batch_unbiased * (cnt - 1) / cnt,
                                   atol=1e-10, rtol=0, err_msg=name)


def test_the_two_padding_and_resize_statements_agree_on_their_own(monkeypatch):
    """the independent helpers against the oracle's, over the shapes the graphs meet"""
    from oracle import np_ops as O
    for size in range(1, 40):
        for k in (1, 3, 5, 7):
            for s in (1, 2):
                for r in (1, 2, 4, 6, 12, 18, 36):
                    if s == 2 and r > 1:
                        continue
                    _, _, (pt, pb, pl, pr) = O.resolve_padding(size, size, k, s, r, 'same')
                    assert (pt, pb) == tf_same_pad(size, k, s, r) == (pl, pr)
    rng = np.random.default_rng(0)
    for f32, tol in ((False, 1e-5), (True, 1e-13)):          # real-valued weights: float32 rounding apart; rounded like TF's kernel: equal
        monkeypatch.setattr(G, 'FLOAT32_COORDS', f32)
        for (h, w, H, W) in [(1, 1, 5, 7), (3, 3, 9, 9), (5, 4, 17, 13), (9, 9, 33, 33), (17, 17, 65, 65), (4, 6, 8, 12), (33, 33, 129, 129)]:
            x = rng.standard_normal((2, h, w, 3))
            ref = np.einsum('oh,nhwc,pw->nopc', resize_matrix(h, H).numpy(), x, resize_matrix(w, W).numpy())
            np.testing.assert_allclose(O.resize_bilinear_fwd(x, H, W), ref, atol=tol, rtol=0)
