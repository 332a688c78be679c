"""A THIRD implementation for the oracle's MobileNetV2 body (TensorFlow cannot be installed here and the reference holds
no arithmetic vectors, so the oracle is pinned by triangulation): HuggingFace `transformers`' MobileNetV2 -- a PyTorch port of the
TF-slim network that HF validates against Google's checkpoints, written by other people from other sources -- is in the image.
With the SAME weights, inference-mode BatchNorm and `tf_padding=True` (TensorFlow's 'SAME' padding, asymmetric on stride 2), its
conv outputs must equal what oracle/np_net.py's restatement of deeplabv3p_mobilenetv2.py:38-199 computes, layer by layer, at
output strides 16 and 8 (the atrous schedule of blocks 6 / 13 on) -- stem conv, depthwise convs, 1x1 convs, BatchNorm (eps 1e-3),
ReLU6, residual adds, block order, channel rounding (make_divisible).  Not the reference itself -- parity stays "unpinned" in the
sense of SURVEY 8c -- but a restatement error in the backbone would have to be made identically by an unrelated code base."""
import numpy as np
import pytest

transformers = pytest.importorskip('transformers')
torch = pytest.importorskip('torch')


def _load(hf, net):
    """oracle parameters (HWIO kernels, Keras BatchNorm names) -> the HF module tree"""
    sd = hf.state_dict()

    def conv(dst, name, depthwise=False):
        w = net.params[name + ('/depthwise_kernel' if depthwise else '/kernel')]
        # HWIO (k, k, cin, cout) -> OIHW; depthwise (k, k, C, 1) -> (C, 1, k, k)
        t = np.transpose(w, (2, 3, 0, 1)) if depthwise else np.transpose(w, (3, 2, 0, 1))
        assert tuple(sd[dst + '.convolution.weight'].shape) == t.shape, (dst, t.shape)
        sd[dst + '.convolution.weight'] = torch.from_numpy(np.ascontiguousarray(t)).double()

    def bn(dst, name):
        for a, b in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'), ('running_var', 'moving_variance')):
            sd[dst + '.normalization.' + a] = torch.from_numpy(net.params[name + '/' + b].copy()).double()
    conv('conv_stem.first_conv', 'Conv'); bn('conv_stem.first_conv', 'Conv_BN')
    conv('conv_stem.conv_3x3', 'expanded_conv_depthwise', True); bn('conv_stem.conv_3x3', 'expanded_conv_depthwise_BN')
    conv('conv_stem.reduce_1x1', 'expanded_conv_project'); bn('conv_stem.reduce_1x1', 'expanded_conv_project_BN')
    for i in range(16):
        p = 'expanded_conv_%d_' % (i + 1)
        conv('layer.%d.expand_1x1' % i, p + 'expand'); bn('layer.%d.expand_1x1' % i, p + 'expand_BN')
        conv('layer.%d.conv_3x3' % i, p + 'depthwise', True); bn('layer.%d.conv_3x3' % i, p + 'depthwise_BN')
        conv('layer.%d.reduce_1x1' % i, p + 'project'); bn('layer.%d.reduce_1x1' % i, p + 'project_BN')
    hf.load_state_dict(sd)


@pytest.mark.parametrize('OS,size', [(16, 65), (16, 97), (8, 65)])
def test_mobilenetv2_body_equals_the_transformers_port(OS, size):
    from oracle.np_net import OracleModel
    o = OracleModel('mobilenetv2', 21, (size, size), OS, dtype=np.float64, seed=3)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (2, size, size, 3))
    o.predict(x)                                   # creates every parameter
    for k, v in o.net.params.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/moving_mean'):
            v[...] = rng.standard_normal(v.shape) * 0.2
        elif k.endswith('/moving_variance'):
            v[...] = rng.uniform(0.5, 2.0, v.shape)
    o.net.record = {}
    o.predict(x)
    rec = o.net.record
    cfg = transformers.MobileNetV2Config(output_stride=OS, tf_padding=True, finegrained_output=True, depth_multiplier=1.0,
                                         hidden_act='relu6', layer_norm_eps=1e-3)
    hf = transformers.MobileNetV2Model(cfg, add_pooling_layer=False).double().eval()
    _load(hf, o.net)
    got = {}
    taps = {'conv_stem.first_conv': 'Conv', 'conv_stem.conv_3x3': 'expanded_conv_depthwise', 'conv_stem.reduce_1x1': 'expanded_conv_project'}
    for i in range(16):
        p = 'expanded_conv_%d_' % (i + 1)
        taps['layer.%d.expand_1x1' % i] = p + 'expand'
        taps['layer.%d.conv_3x3' % i] = p + 'depthwise'
        taps['layer.%d.reduce_1x1' % i] = p + 'project'
    mods = dict(hf.named_modules())
    hooks = [mods[k + '.convolution'].register_forward_hook(lambda m, a, out, name=v: got.__setitem__(name, out.detach().numpy()))
             for k, v in taps.items()]
    with torch.no_grad():
        hf(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy()))
    for h in hooks:
        h.remove()
    assert set(got) == set(taps.values())
    for name in taps.values():
        ref = np.transpose(got[name], (0, 2, 3, 1))
        mine = rec[name]
        assert mine.shape == ref.shape, (name, mine.shape, ref.shape)
        err = float(np.abs(mine - ref).max()) / max(1e-30, float(np.abs(ref).max()))
        assert err < 1e-10, (name, err)
    # (the output stride really is what was asked for: 65 -> 5 at OS 16, 9 at OS 8)
    last = rec['expanded_conv_16_project']
    assert last.shape[1] == (size + OS - 1) // OS and last.shape[-1] == 320


@pytest.mark.parametrize('OS,size,classes', [(16, 65, 21), (8, 97, 19)])
def test_mobilenetv2_lite_logits_equal_the_transformers_deeplab_head(OS, size, classes):
    """the BASELINE configs[0] model (MobileNetV2 + ASPP_Lite_block, layers.py:166-196, + the conv_upsample logits conv,
    model.py:75-79) against transformers' MobileNetV2ForSemanticSegmentation: the image-pooling branch, the 1x1 branch, the concat
    ORDER ([pooled, 1x1]), the projection and the classifier -- same weights, inference mode, logits at the backbone's resolution"""
    from oracle.np_net import OracleModel
    o = OracleModel('mobilenetv2_lite', classes, (size, size), OS, dtype=np.float64, seed=5)
    rng = np.random.default_rng(13)
    x = rng.uniform(-1, 1, (2, size, size, 3))
    o.predict(x)
    for k, v in o.net.params.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/moving_mean') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.2
        elif k.endswith('/moving_variance'):
            v[...] = rng.uniform(0.5, 2.0, v.shape)
    o.net.record = {}
    o.predict(x)
    cfg = transformers.MobileNetV2Config(output_stride=OS, tf_padding=True, finegrained_output=True, depth_multiplier=1.0,
                                         hidden_act='relu6', layer_norm_eps=1e-3, num_labels=classes)
    hf = transformers.MobileNetV2ForSemanticSegmentation(cfg).double().eval()
    _load(hf.mobilenet_v2, o.net)
    sd = hf.state_dict()
    P = o.net.params

    def put(dst, name, bn_name=None):
        sd[dst + '.convolution.weight'] = torch.from_numpy(np.ascontiguousarray(np.transpose(P[name + '/kernel'], (3, 2, 0, 1)))).double()
        if bn_name:
            for a, b in (('weight', 'gamma'), ('bias', 'beta'), ('running_mean', 'moving_mean'), ('running_var', 'moving_variance')):
                sd[dst + '.normalization.' + a] = torch.from_numpy(P[bn_name + '/' + b].copy()).double()
    put('segmentation_head.conv_pool', 'image_pooling', 'image_pooling_BN')
    put('segmentation_head.conv_aspp', 'aspp0', 'aspp0_BN')
    put('segmentation_head.conv_projection', 'concat_projection', 'concat_projection_BN')
    put('segmentation_head.classifier', 'conv_upsample')
    sd['segmentation_head.classifier.convolution.bias'] = torch.from_numpy(P['conv_upsample/bias'].copy()).double()
    hf.load_state_dict(sd)
    with torch.no_grad():
        logits = hf(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy())).logits.numpy()
    ref = np.transpose(logits, (0, 2, 3, 1))
    mine = o.net.record['conv_upsample'][..., :classes]
    assert mine.shape == ref.shape, (mine.shape, ref.shape)
    assert float(np.abs(mine - ref).max()) < 1e-10 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('size', [65, 97])
def test_resnet50_body_equals_the_transformers_port_at_output_stride_32(size):
    """the oracle's restatement of deeplabv3p_resnet50.py:32-139,262-292 with every atrous rate at 1 (OS 32: the plain
    ResNet50 v1 -- stride on the first 1x1 of a stage, conv1_pad + 7x7 stride 2, pool1_pad + 3x3 max pooling) against transformers'
    ResNetModel(downsample_in_bottleneck=True).  The Keras convs carry a bias and BatchNorm eps 1e-3; the port has neither: a bias in
    front of an inference-mode BatchNorm is a shift of its moving mean (exact), eps is set on the modules.  Compared: the output of
    every bottleneck block and the stem (after ReLU), i.e. block order, strides, shortcut convs, padding and pooling."""
    from oracle.np_net import OracleModel
    o = OracleModel('resnet50', 21, (size, size), 32, dtype=np.float64, seed=7)
    rng = np.random.default_rng(17)
    x = rng.uniform(-1, 1, (2, size, size, 3))
    o.predict(x)
    P = o.net.params
    for k, v in P.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/moving_mean') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.2
        elif k.endswith('/moving_variance'):
            v[...] = rng.uniform(0.5, 2.0, v.shape)
    o.net.record = {}
    o.predict(x)
    cfg = transformers.ResNetConfig(downsample_in_bottleneck=True, layer_type='bottleneck', hidden_sizes=[256, 512, 1024, 2048],
                                    depths=[3, 4, 6, 3], embedding_size=64)
    hf = transformers.ResNetModel(cfg).double().eval()
    for m in hf.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    sd = hf.state_dict()

    def put(dst, conv, bn):
        sd[dst + '.convolution.weight'] = torch.from_numpy(np.ascontiguousarray(np.transpose(P[conv + '/kernel'], (3, 2, 0, 1)))).double()
        sd[dst + '.normalization.weight'] = torch.from_numpy(P[bn + '/gamma'].copy())
        sd[dst + '.normalization.bias'] = torch.from_numpy(P[bn + '/beta'].copy())
        sd[dst + '.normalization.running_mean'] = torch.from_numpy(P[bn + '/moving_mean'] - P[conv + '/bias'])
        sd[dst + '.normalization.running_var'] = torch.from_numpy(P[bn + '/moving_variance'].copy())
    put('embedder.embedder', 'conv1', 'bn_conv1')
    blocks = []
    for s, (stage, n) in enumerate([(2, 3), (3, 4), (4, 6), (5, 3)]):
        for b in range(n):
            letter = 'abcdef'[b]
            cn, bn = 'res%d%s_branch' % (stage, letter), 'bn%d%s_branch' % (stage, letter)
            dst = 'encoder.stages.%d.layers.%d' % (s, b)
            for j, part in enumerate(('2a', '2b', '2c')):
                put('%s.layer.%d' % (dst, j), cn + part, bn + part)
            if b == 0:
                put(dst + '.shortcut', cn + '1', bn + '1')
            blocks.append((dst, cn + '2c'))
    hf.load_state_dict(sd)
    # the oracle records raw conv outputs; a block's output is relu(BN(conv 2c) + shortcut): rebuild the NEXT block's input from the
    # record instead -- the first conv of block i+1 reads it -- by comparing the conv outputs the port computes from ITS block outputs
    got = {}
    mods = dict(hf.named_modules())
    hooks = []
    for dst, name in blocks:
        for j, part in enumerate(('2a', '2b', '2c')):
            hooks.append(mods['%s.layer.%d.convolution' % (dst, j)].register_forward_hook(
                lambda m, a, out, key=name[:-2] + part: got.__setitem__(key, out.detach().numpy())))
    hooks.append(mods['embedder.embedder.convolution'].register_forward_hook(lambda m, a, out: got.__setitem__('conv1', out.detach().numpy())))
    with torch.no_grad():
        hf(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy()))
    for h in hooks:
        h.remove()
    assert len(got) == 1 + 3 * 16
    for name, ref in got.items():
        mine = o.net.record[name] - P[name + '/bias']            # the port's conv has no bias (it sits in the moving mean)
        ref = np.transpose(ref, (0, 2, 3, 1))
        assert mine.shape == ref.shape, (name, mine.shape, ref.shape)
        err = float(np.abs(mine - ref).max()) / max(1e-30, float(np.abs(ref).max()))
        assert err < 1e-10, (name, err)


def test_mobilenetv2_lite_training_gradients_equal_the_transformers_port():
    """the TRAINING path of the configs[0] model against the same third party: BatchNorm on batch statistics forward AND backward,
    ReLU6 / ReLU derivatives, depthwise / 1x1 / strided 3x3 data and weight gradients, the residual adds, global average pooling and
    its broadcast back -- torch autograd through transformers' modules on one side, the oracle's hand-written tape on the other.
    Both get the same weights, the same batch and the same gradient at the logits conv (Net.force_grad); every one of the 158
    trainable tensors' gradients must agree (fp64, 1e-7 of the tensor's scale: BatchNorm backward is a difference of sums)."""
    from oracle.np_net import OracleModel
    size, classes, OS, N = 65, 21, 16, 3
    o = OracleModel('mobilenetv2_lite', classes, (size, size), OS, dtype=np.float64, seed=9)
    rng = np.random.default_rng(23)
    x = rng.uniform(-1, 1, (N, size, size, 3))
    o.predict(x)
    P = o.net.params
    for k, v in P.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.2
    cfg = transformers.MobileNetV2Config(output_stride=OS, tf_padding=True, finegrained_output=True, depth_multiplier=1.0,
                                         hidden_act='relu6', layer_norm_eps=1e-3, num_labels=classes, classifier_dropout_prob=0.0)
    hf = transformers.MobileNetV2ForSemanticSegmentation(cfg).double()
    _load(hf.mobilenet_v2, o.net)
    sd = hf.state_dict()

    def put(dst, name, bn_name=None):
        sd[dst + '.convolution.weight'] = torch.from_numpy(np.ascontiguousarray(np.transpose(P[name + '/kernel'], (3, 2, 0, 1)))).double()
        if bn_name:
            sd[dst + '.normalization.weight'] = torch.from_numpy(P[bn_name + '/gamma'].copy())
            sd[dst + '.normalization.bias'] = torch.from_numpy(P[bn_name + '/beta'].copy())
    put('segmentation_head.conv_pool', 'image_pooling', 'image_pooling_BN')
    put('segmentation_head.conv_aspp', 'aspp0', 'aspp0_BN')
    put('segmentation_head.conv_projection', 'concat_projection', 'concat_projection_BN')
    put('segmentation_head.classifier', 'conv_upsample')
    sd['segmentation_head.classifier.convolution.bias'] = torch.from_numpy(P['conv_upsample/bias'].copy())
    hf.load_state_dict(sd)
    hf.train()                                        # BatchNorm on batch statistics (the head's dropout has p = 0)
    logits = hf(torch.from_numpy(np.transpose(x, (0, 3, 1, 2)).copy())).logits
    R = rng.standard_normal(tuple(logits.shape))      # d objective / d logits, NCHW
    (logits * torch.from_numpy(R)).sum().backward()
    # the oracle: same batch, no dropout mask (identity), the backward continued at the logits conv with R
    labels = rng.integers(0, classes, (N, size * size, 1)).astype(np.float64)
    cpad = o.net.params['conv_upsample/kernel'].shape[-1]
    Rn = np.zeros(tuple(np.transpose(R, (0, 2, 3, 1)).shape[:3]) + (cpad,))
    Rn[..., :classes] = np.transpose(R, (0, 2, 3, 1))
    o.net.force_grad = {'conv_upsample': Rn}
    o.net.record = {}
    o.loss_and_grads(x, labels)
    assert float(np.abs(o.net.record['conv_upsample'][..., :classes] - np.transpose(logits.detach().numpy(), (0, 2, 3, 1))).max()) < 1e-9
    grads = o.net.grads
    named = dict(hf.named_parameters())
    pairs = [('mobilenet_v2.conv_stem.first_conv', 'Conv', 'Conv_BN', False), ('mobilenet_v2.conv_stem.conv_3x3', 'expanded_conv_depthwise', 'expanded_conv_depthwise_BN', True),
             ('mobilenet_v2.conv_stem.reduce_1x1', 'expanded_conv_project', 'expanded_conv_project_BN', False)]
    for i in range(16):
        p = 'expanded_conv_%d_' % (i + 1)
        pairs += [('mobilenet_v2.layer.%d.expand_1x1' % i, p + 'expand', p + 'expand_BN', False),
                  ('mobilenet_v2.layer.%d.conv_3x3' % i, p + 'depthwise', p + 'depthwise_BN', True),
                  ('mobilenet_v2.layer.%d.reduce_1x1' % i, p + 'project', p + 'project_BN', False)]
    pairs += [('segmentation_head.conv_pool', 'image_pooling', 'image_pooling_BN', False), ('segmentation_head.conv_aspp', 'aspp0', 'aspp0_BN', False),
              ('segmentation_head.conv_projection', 'concat_projection', 'concat_projection_BN', False),
              ('segmentation_head.classifier', 'conv_upsample', None, False)]
    checked = 0
    for dst, conv, bn, dw in pairs:
        gw = named[dst + '.convolution.weight'].grad.numpy()
        mine = grads[conv + ('/depthwise_kernel' if dw else '/kernel')]
        ref = np.transpose(gw, (2, 3, 0, 1)) if dw else np.transpose(gw, (2, 3, 1, 0))
        if conv == 'conv_upsample':
            mine = mine[..., :classes]
        todo = [(conv + ' kernel', mine, ref)]
        if bn:
            todo += [(bn + ' gamma', grads[bn + '/gamma'], named[dst + '.normalization.weight'].grad.numpy()),
                     (bn + ' beta', grads[bn + '/beta'], named[dst + '.normalization.bias'].grad.numpy())]
        else:
            todo += [(conv + ' bias', grads[conv + '/bias'][:classes], named[dst + '.convolution.bias'].grad.numpy())]
        for what, a, b in todo:
            assert a.shape == b.shape, (what, a.shape, b.shape)
            scale = max(float(np.abs(b).max()), 1e-12)
            # (+ 1e-10 absolute: a beta in front of a conv + BatchNorm pair has an exactly zero gradient -- the next BatchNorm
            # removes the shift -- and both sides then hold 1e-12 of summation noise around it)
            assert float(np.abs(a - b).max()) < 1e-7 * scale + 1e-10, (what, float(np.abs(a - b).max()), scale)
            checked += 1
    assert checked == 3 * 55 - 1        # 54 conv + BatchNorm triples and the classifier's (kernel, bias)
