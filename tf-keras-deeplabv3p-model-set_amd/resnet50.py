"""DeepLabV3+ ResNet50 graph: counterpart of the reference's deeplabv3p/models/deeplabv3p_resnet50.py (identity_block
:32-75, conv_block :78-139, ResNet50 body :142-330 with its output-stride table, Deeplabv3pResNet50 :333-400).
Every DeeplabConv2D here keeps Keras' default use_bias=True in front of its BatchNormalization and is he_normal
initialised; BatchNormalization uses the Keras defaults (eps 1e-3, momentum 0.99)."""
from .graph import GraphBuilder
from .layers import ASPP_block, Decoder_block


def _conv_bn(g, x, filters, k, conv_name, bn_name, stride=1, rate=1, padding='same'):
    x = g.conv2d(x, filters, k, conv_name, stride=stride, rate=rate, padding=padding, use_bias=True,
                 kernel_initializer='he_normal')
    return g.batchnorm(x, bn_name)


def identity_block(g, input_tensor, kernel_size, filters, stage, block, rate=1):
    f1, f2, f3 = filters
    cn, bn = 'res' + str(stage) + block + '_branch', 'bn' + str(stage) + block + '_branch'
    x = _conv_bn(g, input_tensor, f1, 1, cn + '2a', bn + '2a')
    x = g.relu(x)
    x = _conv_bn(g, x, f2, kernel_size, cn + '2b', bn + '2b', rate=rate)
    x = g.relu(x)
    x = _conv_bn(g, x, f3, 1, cn + '2c', bn + '2c')
    x = g.add(input_tensor, x, keras_inputs=[x, input_tensor])      # add([x, input_tensor]) (:75)
    return g.relu(x)


def conv_block(g, input_tensor, kernel_size, filters, stage, block, strides=2, rate=1):
    f1, f2, f3 = filters
    cn, bn = 'res' + str(stage) + block + '_branch', 'bn' + str(stage) + block + '_branch'
    x = _conv_bn(g, input_tensor, f1, 1, cn + '2a', bn + '2a', stride=strides, padding='valid')
    x = g.relu(x)
    x = _conv_bn(g, x, f2, kernel_size, cn + '2b', bn + '2b', rate=rate)
    x = g.relu(x)
    x = _conv_bn(g, x, f3, 1, cn + '2c', bn + '2c')
    shortcut = _conv_bn(g, input_tensor, f3, 1, cn + '1', bn + '1', stride=strides, padding='valid')
    x = g.add(shortcut, x, keras_inputs=[x, shortcut])               # add([x, shortcut]) (:140)
    return g.relu(x)


def ResNet50_body(g, input_tensor, OS):
    if OS == 8:
        s16, r16, s32, r32 = 1, 2, 1, 4
    elif OS == 16:
        s16, r16, s32, r32 = 2, 1, 1, 2
    elif OS == 32:
        s16, r16, s32, r32 = 2, 1, 2, 1
    else:
        raise ValueError('invalid output stride', OS)
    x = _conv_bn(g, input_tensor, 64, 7, 'conv1', 'bn_conv1', stride=2, padding=(3, 3, 3, 3))     # conv1_pad + valid
    x = g.relu(x)
    x = g.maxpool2d(x, 3, 2, (1, 1, 1, 1), pad_name='pool1_pad')
    x = conv_block(g, x, 3, [64, 64, 256], stage=2, block='a', strides=1)
    x = identity_block(g, x, 3, [64, 64, 256], stage=2, block='b')
    x = identity_block(g, x, 3, [64, 64, 256], stage=2, block='c')
    skip = x
    x = conv_block(g, x, 3, [128, 128, 512], stage=3, block='a')
    for b in 'bcd':
        x = identity_block(g, x, 3, [128, 128, 512], stage=3, block=b)
    x = conv_block(g, x, 3, [256, 256, 1024], stage=4, block='a', strides=s16)
    for b in 'bcdef':
        x = identity_block(g, x, 3, [256, 256, 1024], stage=4, block=b, rate=r16)
    x = conv_block(g, x, 3, [512, 512, 2048], stage=5, block='a', strides=s32, rate=r16)
    for b in 'bc':
        x = identity_block(g, x, 3, [512, 512, 2048], stage=5, block=b, rate=r32)
    return x, skip, len(g.layers)


def Deeplabv3pResNet50(input_shape=(512, 512, 3), weights=None, input_tensor=None, num_classes=21, OS=8, seed=0):
    """ResNet50 + ASPP + decoder; returns (graph, head_input, backbone_len) like the other builders"""
    if weights not in {'imagenet', None}:
        raise ValueError('The `weights` argument should be either `imagenet` (pre-trained on Imagenet) or '
                         '`None` (random initialization)')
    g = input_tensor if isinstance(input_tensor, GraphBuilder) else GraphBuilder(input_shape, 'deeplabv3p_resnet50', seed)
    x, skip_feature, backbone_len = ResNet50_body(g, g.input, OS)
    g.tap('backbone_out', x)
    x = ASPP_block(g, x, OS)
    x = Decoder_block(g, x, skip_feature)
    return g, x, backbone_len
