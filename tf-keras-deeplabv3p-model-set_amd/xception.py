"""DeepLabV3+ (modified aligned) Xception graph: counterpart of the reference's
deeplabv3p/models/deeplabv3p_xception.py (_conv2d_same :25-54, _xception_block :57-93, Xception_body
:96-163, Deeplabv3pXception :167-238)."""
from .graph import GraphBuilder
from .layers import SepConv_BN, ASPP_block, Decoder_block


def _conv2d_same(g, x, filters, prefix, stride=1, kernel_size=3, rate=1):
    """'same' padding that does not drift by one pixel at stride 2 (reference :25-54)"""
    if stride == 1:
        return g.conv2d(x, filters, kernel_size, prefix, stride=1, rate=rate, padding='same')
    kernel_size_effective = kernel_size + (kernel_size - 1) * (rate - 1)
    pad_total = kernel_size_effective - 1
    pad_beg = pad_total // 2
    pad_end = pad_total - pad_beg
    return g.conv2d(x, filters, kernel_size, prefix, stride=stride, rate=rate,
                    padding=(pad_beg, pad_end, pad_beg, pad_end))


def _xception_block(g, inputs, depth_list, prefix, skip_connection_type, stride, rate=1, depth_activation=False,
                    return_skip=False):
    """three SepConv_BN + conv / sum / no shortcut (reference :57-93)"""
    residual = inputs
    skip = None
    for i in range(3):
        residual = SepConv_BN(g, residual, depth_list[i], prefix + '_separable_conv{}'.format(i + 1),
                              stride=stride if i == 2 else 1, rate=rate, depth_activation=depth_activation)
        if i == 1:
            skip = residual
            if return_skip:
                # the decoder consumes this BN output as is while the third SepConv applies ReLU to it:
                # two different activations on one BatchNormalization -> materialise it once
                skip = g.materialize(residual, name=prefix + '_skip')
                residual = skip
    if skip_connection_type == 'conv':
        shortcut = _conv2d_same(g, inputs, depth_list[-1], prefix + '_shortcut', kernel_size=1, stride=stride)
        shortcut = g.batchnorm(shortcut, prefix + '_shortcut_BN')
        outputs = g.add(residual, shortcut)
    elif skip_connection_type == 'sum':
        outputs = g.add(inputs, residual, keras_inputs=[residual, inputs])       # add([residual, inputs]) (:87)
    elif skip_connection_type == 'none':
        outputs = residual
    if return_skip:
        return outputs, skip
    return outputs


def Xception_body(g, input_tensor, OS):
    if OS == 8:
        s16, r16, s32, r32 = 1, 2, 1, 4
    elif OS == 16:
        s16, r16, s32, r32 = 2, 1, 1, 2
    elif OS == 32:
        s16, r16, s32, r32 = 2, 1, 2, 1
    else:
        raise ValueError('invalid output stride', OS)
    x = g.conv2d(input_tensor, 32, 3, 'entry_flow_conv1_1', stride=2, padding='same')
    x = g.batchnorm(x, 'entry_flow_conv1_1_BN')
    x = g.relu(x)
    x = _conv2d_same(g, x, 64, 'entry_flow_conv1_2', kernel_size=3, stride=1)
    x = g.batchnorm(x, 'entry_flow_conv1_2_BN')
    x = g.relu(x)
    x = _xception_block(g, x, [128, 128, 128], 'entry_flow_block1', 'conv', 2)
    x, skip = _xception_block(g, x, [256, 256, 256], 'entry_flow_block2', 'conv', 2, return_skip=True)
    x = _xception_block(g, x, [728, 728, 728], 'entry_flow_block3', 'conv', s16)
    for i in range(16):
        x = _xception_block(g, x, [728, 728, 728], 'middle_flow_unit_{}'.format(i + 1), 'sum', 1, rate=r16)
    x = _xception_block(g, x, [728, 1024, 1024], 'exit_flow_block1', 'conv', s32, rate=r16)
    x = _xception_block(g, x, [1536, 1536, 2048], 'exit_flow_block2', 'none', 1, rate=r32, depth_activation=True)
    return x, skip, len(g.layers)


def Deeplabv3pXception(input_shape=(512, 512, 3), weights=None, input_tensor=None, num_classes=21, OS=16, seed=0):
    """Xception + ASPP + decoder; returns (graph, head_input, backbone_len) like the MobileNetV2 builders"""
    if weights not in {'pascalvoc', None}:
        raise ValueError('The `weights` argument should be either `None` (random initialization) or `pascalvoc` '
                         '(pre-trained on PASCAL VOC)')
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_xception', seed)
    x, skip_feature, backbone_len = Xception_body(g, g.input, OS)
    g.tap('backbone_out', x)
    x = ASPP_block(g, x, OS)
    x = Decoder_block(g, x, skip_feature)
    return g, x, backbone_len
