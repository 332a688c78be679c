"""DeepLabV3+ MobileNetV3 graphs: counterpart of the reference's deeplabv3p/models/deeplabv3p_mobilenetv3.py
(correct_pad :50-72, hard_sigmoid / hard_swish :98-103, _depth :112-119, _se_block :122-146,
_inverted_res_block :149-201, MobileNetV3 stem :343-355, MobileNetV3Small.stack_fn :469-499,
MobileNetV3Large.stack_fn :551-593, Deeplabv3pMobileNetV3Large :615-681, Deeplabv3pLiteMobileNetV3Large :684-751,
Deeplabv3pMobileNetV3Small :754-820, Deeplabv3pLiteMobileNetV3Small :823-890)."""
from .graph import GraphBuilder, ACT_RELU, ACT_HSWISH, ACT_HSIGMOID
from .layers import ASPP_block, ASPP_Lite_block, Decoder_block


def _depth(v, divisor=8, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def correct_pad(H, W, kernel_size):
    """zero-padding for a stride-2 'valid' conv that matches 'same' at odd sizes (reference :50-72)"""
    adjust = (1 - H % 2, 1 - W % 2)
    correct = (kernel_size // 2, kernel_size // 2)
    return (correct[0] - adjust[0], correct[0], correct[1] - adjust[1], correct[1])


def _act(g, x, activation):
    return g.activation(x, activation, kind='Activation' if activation != ACT_RELU else 'ReLU')


def _se_block(g, inputs, filters, se_ratio, prefix):
    x = g.global_avgpool(inputs, prefix + 'squeeze_excite/AvgPool', kind='GlobalAveragePooling2D')
    x = g.passthrough(x, 'Reshape', (1, 1, filters))
    x = g.conv2d(x, _depth(filters * se_ratio), 1, prefix + 'squeeze_excite/Conv', use_bias=True)
    x = g.relu(x, prefix + 'squeeze_excite/Relu')
    x = g.conv2d(x, filters, 1, prefix + 'squeeze_excite/Conv_1', use_bias=True)
    x = g.activation(x, ACT_HSIGMOID, kind='Activation')
    return g.se_multiply(inputs, x, prefix + 'squeeze_excite/Mul')


def _inverted_res_block(g, x, expansion, filters, kernel_size, stride, se_ratio, activation, block_id,
                        skip_connection=False, rate=1):
    shortcut = x
    prefix = 'expanded_conv/'
    infilters = x.shape[2]
    if block_id:
        prefix = 'expanded_conv_{}/'.format(block_id)
        x = g.conv2d(x, _depth(infilters * expansion), 1, prefix + 'expand')
        x = g.batchnorm(x, prefix + 'expand/BatchNorm', eps=1e-3, momentum=0.999)
        x = _act(g, x, activation)
    x = g.dwconv2d(x, kernel_size, prefix + 'depthwise/Conv', stride=stride, rate=rate, padding='same')
    x = g.batchnorm(x, prefix + 'depthwise/BatchNorm', eps=1e-3, momentum=0.999)
    x = _act(g, x, activation)
    if se_ratio:
        x = _se_block(g, x, _depth(infilters * expansion), se_ratio, prefix)
    x = g.conv2d(x, filters, 1, prefix + 'project')
    x = g.batchnorm(x, prefix + 'project/BatchNorm', eps=1e-3, momentum=0.999)
    if skip_connection:
        x = g.add(shortcut, x, prefix + 'Add')
    return x


def MobileNetV3Large_body(g, input_tensor, OS, alpha=1.0):
    if OS == 8:
        s16, r16, s32, r32 = 1, 2, 1, 4
    elif OS == 16:
        s16, r16, s32, r32 = 2, 1, 1, 2
    elif OS == 32:
        s16, r16, s32, r32 = 2, 1, 2, 1
    else:
        raise ValueError('invalid output stride', OS)
    H, W, _ = input_tensor.shape
    kernel, activation, se_ratio = 5, ACT_HSWISH, 0.25
    x = g.conv2d(input_tensor, 16, 3, 'Conv', stride=2, padding=correct_pad(H, W, 3))
    x = g.batchnorm(x, 'Conv/BatchNorm', eps=1e-3, momentum=0.999)
    x = _act(g, x, activation)
    d = lambda v: _depth(v * alpha)
    blk = lambda x, **kw: _inverted_res_block(g, x, **kw)
    RE = ACT_RELU
    x = blk(x, expansion=1, filters=d(16), kernel_size=3, stride=1, se_ratio=None, activation=RE, block_id=0, skip_connection=True)
    x = blk(x, expansion=4, filters=d(24), kernel_size=3, stride=2, se_ratio=None, activation=RE, block_id=1)
    x = blk(x, expansion=3, filters=d(24), kernel_size=3, stride=1, se_ratio=None, activation=RE, block_id=2, skip_connection=True)
    skip = x
    x = blk(x, expansion=3, filters=d(40), kernel_size=kernel, stride=2, se_ratio=se_ratio, activation=RE, block_id=3)
    x = blk(x, expansion=3, filters=d(40), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=RE, block_id=4, skip_connection=True)
    x = blk(x, expansion=3, filters=d(40), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=RE, block_id=5, skip_connection=True)
    x = blk(x, expansion=6, filters=d(80), kernel_size=3, stride=s16, se_ratio=None, activation=activation, block_id=6)
    x = blk(x, expansion=2.5, filters=d(80), kernel_size=3, stride=1, se_ratio=None, activation=activation, block_id=7, skip_connection=True, rate=r16)
    x = blk(x, expansion=2.3, filters=d(80), kernel_size=3, stride=1, se_ratio=None, activation=activation, block_id=8, skip_connection=True, rate=r16)
    x = blk(x, expansion=2.3, filters=d(80), kernel_size=3, stride=1, se_ratio=None, activation=activation, block_id=9, skip_connection=True, rate=r16)
    x = blk(x, expansion=6, filters=d(112), kernel_size=3, stride=1, se_ratio=se_ratio, activation=activation, block_id=10, rate=r16)
    x = blk(x, expansion=6, filters=d(112), kernel_size=3, stride=1, se_ratio=se_ratio, activation=activation, block_id=11, skip_connection=True, rate=r16)
    x = blk(x, expansion=6, filters=d(160), kernel_size=kernel, stride=s32, se_ratio=se_ratio, activation=activation, block_id=12, rate=r16)
    x = blk(x, expansion=6, filters=d(160), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=13, skip_connection=True, rate=r32)
    x = blk(x, expansion=6, filters=d(160), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=14, skip_connection=True, rate=r32)
    # the reference's Conv_1 / Conv_1/BatchNorm tail only hosts the ImageNet weight file and is not part of
    # the returned graph
    return x, skip, len(g.layers)


def Deeplabv3pMobileNetV3Large(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None, num_classes=21,
                               OS=8, seed=0):
    if weights not in {'imagenet', None}:
        raise ValueError('The `weights` argument should be either `imagenet` (pre-trained on Imagenet) or '
                         '`None` (random initialization)')
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv3large', seed)
    x, skip_feature, backbone_len = MobileNetV3Large_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_block(g, x, OS)
    x = Decoder_block(g, x, skip_feature)
    return g, x, backbone_len


def _os_table(OS):
    if OS == 8:
        return 1, 2, 1, 4
    if OS == 16:
        return 2, 1, 1, 2
    if OS == 32:
        return 2, 1, 2, 1
    raise ValueError('invalid output stride', OS)


def MobileNetV3Small_body(g, input_tensor, OS, alpha=1.0):
    """stem (:343-355) + MobileNetV3Small.stack_fn (:469-499); the skip feature is block 0's output (1/4 resolution,
    16 channels)"""
    s16, r16, s32, r32 = _os_table(OS)
    H, W, _ = input_tensor.shape
    kernel, activation, se_ratio = 5, ACT_HSWISH, 0.25
    x = g.conv2d(input_tensor, 16, 3, 'Conv', stride=2, padding=correct_pad(H, W, 3))
    x = g.batchnorm(x, 'Conv/BatchNorm', eps=1e-3, momentum=0.999)
    x = _act(g, x, activation)
    d = lambda v: _depth(v * alpha)
    blk = lambda x, **kw: _inverted_res_block(g, x, **kw)
    RE = ACT_RELU
    x = blk(x, expansion=1, filters=d(16), kernel_size=3, stride=2, se_ratio=se_ratio, activation=RE, block_id=0)
    skip = x
    x = blk(x, expansion=72. / 16, filters=d(24), kernel_size=3, stride=2, se_ratio=None, activation=RE, block_id=1)
    x = blk(x, expansion=88. / 24, filters=d(24), kernel_size=3, stride=1, se_ratio=None, activation=RE, block_id=2, skip_connection=True)
    x = blk(x, expansion=4, filters=d(40), kernel_size=kernel, stride=s16, se_ratio=se_ratio, activation=activation, block_id=3)
    x = blk(x, expansion=6, filters=d(40), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=4, skip_connection=True, rate=r16)
    x = blk(x, expansion=6, filters=d(40), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=5, skip_connection=True, rate=r16)
    x = blk(x, expansion=3, filters=d(48), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=6, rate=r16)
    x = blk(x, expansion=3, filters=d(48), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=7, skip_connection=True, rate=r16)
    x = blk(x, expansion=6, filters=d(96), kernel_size=kernel, stride=s32, se_ratio=se_ratio, activation=activation, block_id=8, rate=r16)
    x = blk(x, expansion=6, filters=d(96), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=9, skip_connection=True, rate=r32)
    x = blk(x, expansion=6, filters=d(96), kernel_size=kernel, stride=1, se_ratio=se_ratio, activation=activation, block_id=10, skip_connection=True, rate=r32)
    return x, skip, len(g.layers)


def _check_weights(weights):
    if weights not in {'imagenet', None}:
        raise ValueError('The `weights` argument should be either `imagenet` (pre-trained on Imagenet) or '
                         '`None` (random initialization)')


def Deeplabv3pLiteMobileNetV3Large(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None,
                                   num_classes=21, OS=8, seed=0):
    """MobileNetV3-Large + ASPP-Lite, no decoder (:684-751)"""
    _check_weights(weights)
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv3large_lite', seed)
    x, _, backbone_len = MobileNetV3Large_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_Lite_block(g, x)
    return g, x, backbone_len


def Deeplabv3pMobileNetV3Small(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None, num_classes=21,
                               OS=8, seed=0):
    """MobileNetV3-Small + ASPP + decoder (:754-820)"""
    _check_weights(weights)
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv3small', seed)
    x, skip_feature, backbone_len = MobileNetV3Small_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_block(g, x, OS)
    x = Decoder_block(g, x, skip_feature)
    return g, x, backbone_len


def Deeplabv3pLiteMobileNetV3Small(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None,
                                   num_classes=21, OS=8, seed=0):
    """MobileNetV3-Small + ASPP-Lite, no decoder (:823-890)"""
    _check_weights(weights)
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv3small_lite', seed)
    x, _, backbone_len = MobileNetV3Small_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_Lite_block(g, x)
    return g, x, backbone_len

