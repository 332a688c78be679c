"""DeepLabV3+ MobileNetV2 graphs: counterparts of the reference's
deeplabv3p/models/deeplabv3p_mobilenetv2.py (_make_divisible :28, _inverted_res_block :38-74,
MobileNetV2_body :77-199, Deeplabv3pMobileNetV2 :202-270, Deeplabv3pLiteMobileNetV2 :273-351)."""
from .graph import GraphBuilder
from .layers import ASPP_block, ASPP_Lite_block, Decoder_block


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _inverted_res_block(g, inputs, expansion, stride, alpha, filters, block_id, skip_connection, rate=1):
    in_channels = inputs.shape[2]
    pointwise_filters = _make_divisible(int(filters * alpha), 8)
    x = inputs
    prefix = 'expanded_conv_{}_'.format(block_id)
    if block_id:
        x = g.conv2d(x, expansion * in_channels, 1, prefix + 'expand')
        x = g.batchnorm(x, prefix + 'expand_BN', eps=1e-3, momentum=0.999)
        x = g.relu6(x)
    else:
        prefix = 'expanded_conv_'
    x = g.dwconv2d(x, 3, prefix + 'depthwise', stride=stride, rate=rate, padding='same')
    x = g.batchnorm(x, prefix + 'depthwise_BN', eps=1e-3, momentum=0.999)
    x = g.relu6(x, prefix + 'depthwise_relu')
    x = g.conv2d(x, pointwise_filters, 1, prefix + 'project')
    x = g.batchnorm(x, prefix + 'project_BN', eps=1e-3, momentum=0.999)
    if skip_connection:
        return g.add(inputs, x, prefix + 'add')
    return x


def MobileNetV2_body(g, input_tensor, OS, alpha):
    """MobileNetV2 feature extractor with output-stride control.  The reference also builds
    Conv_1/Conv_1_bn/out_relu only to host the ImageNet weight file (:166-194); that branch is not
    part of the returned graph and is not built."""
    if OS == 8:
        s16, r16, s32, r32 = 1, 2, 1, 4
    elif OS == 16:
        s16, r16, s32, r32 = 2, 1, 1, 2
    elif OS == 32:
        s16, r16, s32, r32 = 2, 1, 2, 1
    else:
        raise ValueError('invalid output stride', OS)
    first_block_filters = _make_divisible(32 * alpha, 8)
    x = g.conv2d(input_tensor, first_block_filters, 3, 'Conv', stride=2, padding='same')
    x = g.batchnorm(x, 'Conv_BN', eps=1e-3, momentum=0.999)
    x = g.relu6(x)
    blk = lambda x, **kw: _inverted_res_block(g, x, alpha=alpha, **kw)
    x = blk(x, filters=16, stride=1, expansion=1, block_id=0, skip_connection=False)
    x = blk(x, filters=24, stride=2, expansion=6, block_id=1, skip_connection=False)
    x = blk(x, filters=24, stride=1, expansion=6, block_id=2, skip_connection=True)
    skip = x
    x = blk(x, filters=32, stride=2, expansion=6, block_id=3, skip_connection=False)
    x = blk(x, filters=32, stride=1, expansion=6, block_id=4, skip_connection=True)
    x = blk(x, filters=32, stride=1, expansion=6, block_id=5, skip_connection=True)
    x = blk(x, filters=64, stride=s16, expansion=6, block_id=6, skip_connection=False)
    x = blk(x, filters=64, stride=1, rate=r16, expansion=6, block_id=7, skip_connection=True)
    x = blk(x, filters=64, stride=1, rate=r16, expansion=6, block_id=8, skip_connection=True)
    x = blk(x, filters=64, stride=1, rate=r16, expansion=6, block_id=9, skip_connection=True)
    x = blk(x, filters=96, stride=1, rate=r16, expansion=6, block_id=10, skip_connection=False)
    x = blk(x, filters=96, stride=1, rate=r16, expansion=6, block_id=11, skip_connection=True)
    x = blk(x, filters=96, stride=1, rate=r16, expansion=6, block_id=12, skip_connection=True)
    x = blk(x, filters=160, stride=s32, rate=r16, expansion=6, block_id=13, skip_connection=False)
    x = blk(x, filters=160, stride=1, rate=r32, expansion=6, block_id=14, skip_connection=True)
    x = blk(x, filters=160, stride=1, rate=r32, expansion=6, block_id=15, skip_connection=True)
    x = blk(x, filters=320, stride=1, rate=r32, expansion=6, block_id=16, skip_connection=False)
    backbone_len = len(g.layers)
    return x, skip, backbone_len


def Deeplabv3pMobileNetV2(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None, num_classes=21,
                          OS=8, seed=0):
    """MobileNetV2 + ASPP + decoder.  Returns (graph, head_input, backbone_len): the 21-class stub head of
    the reference (:255-258) is dropped again by get_deeplabv3p_model (model.py:65), so the graph ends
    at the tensor that feeds it."""
    if weights not in {'imagenet', None}:
        raise ValueError('The `weights` argument should be either `imagenet` (pre-trained on Imagenet) or '
                         '`None` (random initialization)')
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv2', seed)
    x, skip_feature, backbone_len = MobileNetV2_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_block(g, x, OS)
    g.tap('aspp_out', x)
    x = Decoder_block(g, x, skip_feature)
    return g, x, backbone_len


def Deeplabv3pLiteMobileNetV2(input_shape=(512, 512, 3), alpha=1.0, weights=None, input_tensor=None,
                              num_classes=21, OS=8, seed=0):
    """MobileNetV2 + ASPP-Lite, no decoder"""
    if weights not in {'pascalvoc', 'imagenet', None}:
        raise ValueError('The `weights` argument should be either `pascalvoc` (pre-trained on PASCAL VOC) '
                         '`imagenet` (pre-trained on Imagenet) or `None` (random initialization)')
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_mobilenetv2_lite', seed)
    x, _, backbone_len = MobileNetV2_body(g, g.input, OS, alpha)
    g.tap('backbone_out', x)
    x = ASPP_Lite_block(g, x)
    return g, x, backbone_len
