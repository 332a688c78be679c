"""The HIP path against a THIRD PARTY directly, no oracle in between: the BASELINE configs[0] model
(get_deeplabv3p_model('mobilenetv2_lite'): MobileNetV2 at output stride 16 + ASPP_Lite_block + conv_upsample + pred_resize + Softmax,
deeplabv3p/model.py:51-117) next to HuggingFace transformers' MobileNetV2ForSemanticSegmentation -- a PyTorch port of the TF-slim
network written by other people -- with the SAME weights, run in float64 on the CPU under torch autograd.

  * inference: class probabilities at full resolution (the port's logits upsampled with torch's bilinear, align_corners=False
    = tf.image.resize in TF2, then softmax);
  * training: one step's loss (Keras' reduction: sum over the labelled pixels / ALL pixels, deeplabv3p/loss.py:121-156) and every
    parameter gradient, BatchNorm on batch statistics, the device's own dropout mask handed to the port.

TensorFlow cannot be installed here, so this is the closest thing to "the product against somebody else's implementation of the
reference's network" the image offers; tests/test_oracle_vs_transformers.py is the same comparison for the oracle (1e-10 / 1e-7)."""
import numpy as np
import pytest

from conftest import load_pkg

transformers = pytest.importorskip('transformers')
torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _port(m_weights, classes, OS, dropout_mask=None):
    from test_oracle_vs_transformers import _load

    class _P:       # what _load expects: an object with .params
        params = m_weights
    cfg = transformers.MobileNetV2Config(output_stride=OS, tf_padding=True, finegrained_output=True, depth_multiplier=1.0,
                                         hidden_act='relu6', layer_norm_eps=1e-3, num_labels=classes, classifier_dropout_prob=0.0)
    hf = transformers.MobileNetV2ForSemanticSegmentation(cfg).double()
    _load(hf.mobilenet_v2, _P)
    sd = hf.state_dict()
    P = m_weights

    def put(dst, name, bn_name=None):
        sd[dst + '.convolution.weight'] = torch.from_numpy(np.ascontiguousarray(np.transpose(P[name + '/kernel'], (3, 2, 0, 1)))).double()
        if bn_name:
            for a, b in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'), ('running_var', 'moving_variance')):
                sd[dst + '.normalization.' + a] = torch.from_numpy(np.asarray(P[bn_name + '/' + b], np.float64).copy())
    put('segmentation_head.conv_pool', 'image_pooling', 'image_pooling_BN')
    put('segmentation_head.conv_aspp', 'aspp0', 'aspp0_BN')
    put('segmentation_head.conv_projection', 'concat_projection', 'concat_projection_BN')
    put('segmentation_head.classifier', 'conv_upsample')
    sd['segmentation_head.classifier.convolution.bias'] = torch.from_numpy(np.asarray(P['conv_upsample/bias'], np.float64).copy())
    hf.load_state_dict(sd)
    if dropout_mask is not None:
        keep = torch.from_numpy(np.transpose(dropout_mask, (0, 3, 1, 2)).astype(np.float64))

        class Keep(torch.nn.Module):            # Keras Dropout(0.5) with the DEVICE's keep mask (layers.py:194)
            def forward(self, t):
                return t * keep * 2.0
        hf.segmentation_head.dropout = Keep()
    return hf


def _weights(pkg, model_type, classes, size, OS, training):
    m = pkg.get_deeplabv3p_model(model_type, classes, (size, size), OS, training=training)
    w = {k: np.asarray(v, np.float64).copy() for k, v in m.get_weights_by_name().items()}
    rng = np.random.default_rng(31)
    for k, v in w.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/moving_mean') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.1
        elif k.endswith('/moving_variance'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
    w = {k: v.astype(np.float32).astype(np.float64) for k, v in w.items()}      # exactly what the device will hold
    m.set_weights_by_name({k: v.astype(np.float32) for k, v in w.items()})
    return m, w


@pytest.mark.parametrize('size,OS', [(65, 16), (97, 16), (97, 8), (129, 8)])
def test_predict_equals_the_transformers_port(size, OS):
    pkg = load_pkg()
    classes, N = 21, 2
    m, w = _weights(pkg, 'mobilenetv2_lite', classes, size, OS, training=False)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (N, size, size, 3)).astype(np.float32)
    p = m.predict(x)
    hf = _port(w, classes, OS).eval()
    with torch.no_grad():
        logits = hf(torch.from_numpy(np.transpose(x.astype(np.float64), (0, 3, 1, 2)).copy())).logits
        up = torch.nn.functional.interpolate(logits, size=(size, size), mode='bilinear', align_corners=False)
        ref = torch.softmax(up, 1).permute(0, 2, 3, 1).numpy()
    assert p.shape == ref.shape
    err = float(np.abs(p - ref).max())
    assert err < 2e-5, err          # (measured: 1e-6 .. 3e-6)


@pytest.mark.parametrize('size', [65, pytest.param(129, marks=pytest.mark.release)])
def test_train_step_loss_and_gradients_equal_the_transformers_port(size):
    pkg = load_pkg()
    classes, OS, N = 21, 16, 4
    m, w = _weights(pkg, 'mobilenetv2_lite', classes, size, OS, training=True)
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    m.use_graphs = False
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (N, size, size, 3)).astype(np.float32)
    y = rng.integers(0, classes, (N, size * size, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    loss = m.train_on_batch(x, y)
    ex = m._executor(N, True)
    drop = [op for op in m.graph.ops if op.kind == 'materialize' and op.rate > 0][0]
    mask = ex.dropout_mask(drop).cpu().numpy()
    hf = _port(w, classes, OS, dropout_mask=mask).train()
    logits = hf(torch.from_numpy(np.transpose(x.astype(np.float64), (0, 3, 1, 2)).copy())).logits
    up = torch.nn.functional.interpolate(logits, size=(size, size), mode='bilinear', align_corners=False)
    lab = torch.from_numpy(y.reshape(N, size, size).astype(np.int64))
    valid = lab != 255
    logp = torch.log_softmax(up, 1).gather(1, lab.clamp(0, classes - 1).unsqueeze(1)).squeeze(1)
    ce = -(logp * valid).sum() / (N * size * size)            # Keras: the mean runs over ALL pixels, ignored ones included
    ce.backward()
    assert abs(loss - float(ce.detach())) < 2e-5 * max(1.0, abs(float(ce.detach()))), (loss, float(ce.detach()))
    named = dict(hf.named_parameters())
    st = m._store
    byname = {p.name: p for p in m.graph.all_params()}
    pairs = [('mobilenet_v2.conv_stem.first_conv', 'Conv', 'Conv_BN', False), ('mobilenet_v2.conv_stem.conv_3x3', 'expanded_conv_depthwise', 'expanded_conv_depthwise_BN', True),
             ('mobilenet_v2.conv_stem.reduce_1x1', 'expanded_conv_project', 'expanded_conv_project_BN', False)]
    for i in range(16):
        pfx = 'expanded_conv_%d_' % (i + 1)
        pairs += [('mobilenet_v2.layer.%d.expand_1x1' % i, pfx + 'expand', pfx + 'expand_BN', False),
                  ('mobilenet_v2.layer.%d.conv_3x3' % i, pfx + 'depthwise', pfx + 'depthwise_BN', True),
                  ('mobilenet_v2.layer.%d.reduce_1x1' % i, pfx + 'project', pfx + 'project_BN', False)]
    pairs += [('segmentation_head.conv_pool', 'image_pooling', 'image_pooling_BN', False), ('segmentation_head.conv_aspp', 'aspp0', 'aspp0_BN', False),
              ('segmentation_head.conv_projection', 'concat_projection', 'concat_projection_BN', False),
              ('segmentation_head.classifier', 'conv_upsample', None, False)]
    num = den = 0.0
    worst = ('', 0.0)
    checked = 0
    for dst, conv, bn, dw in pairs:
        gw = named[dst + '.convolution.weight'].grad.numpy()
        todo = [(conv + ('/depthwise_kernel' if dw else '/kernel'), np.transpose(gw, (2, 3, 0, 1)) if dw else np.transpose(gw, (2, 3, 1, 0)))]
        if bn:
            todo += [(bn + '/gamma', named[dst + '.normalization.weight'].grad.numpy()), (bn + '/beta', named[dst + '.normalization.bias'].grad.numpy())]
        else:
            todo += [(conv + '/bias', named[dst + '.convolution.bias'].grad.numpy())]
        for name, ref in todo:
            g = np.asarray(st.get(byname[name], st.G), np.float64)
            g = g[..., :ref.shape[-1]] if g.shape != ref.shape else g          # the class dimension is padded on the device
            assert g.shape == ref.shape, (name, g.shape, ref.shape)
            checked += 1
            num += float(((g - ref) ** 2).sum()); den += float((ref ** 2).sum())
            if float(np.abs(ref).max()) < 1e-7:          # a beta in front of a conv + BatchNorm pair: exactly zero, both hold noise
                assert float(np.abs(g).max()) < 1e-5, name
                continue
            r = float(np.abs(g - ref).max() / np.abs(ref).max())
            if r > worst[1]:
                worst = (name, r)
    assert checked == 3 * 55 - 1
    # fp32 on the device against fp64 in the port, through 52 BatchNorms on batch statistics: a pre-activation at rounding distance
    # of a ReLU6 kink takes the other branch in fp32 -- an O(1) change of one of a channel's few hundred terms
    # (tests/test_model_gpu.py injects the device's branch pattern into the oracle for that reason and then holds 8e-3 per tensor;
    # nothing is injected into the third party here, so the bound is the un-injected one: measured 1.1e-2 in relative L2 over all
    # gradients at 65 x 65 -- 5 x 5 maps, 100 samples per channel)
    import json, os
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'product_vs_transformers.jsonl'), 'a') as f:
            f.write(json.dumps(dict(size=size, loss=loss, port_loss=float(ce.detach()), relative_l2=float(np.sqrt(num / den)), worst=worst)) + '\n')
    except OSError:
        pass
    assert np.sqrt(num / den) < 3e-2, np.sqrt(num / den)
    assert worst[1] < 0.25, worst
