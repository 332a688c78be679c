"""An independent torch.nn.Module statement of the DeepLabV3+ graphs that have no third-party port in this image (transformers
carries MobileNetV2 and the plain ResNet only -- tests/test_oracle_vs_transformers.py): the modified aligned Xception body, the
MobileNetV3 Large / Small bodies with squeeze-excite, ResNet50 WITH its atrous schedule (output strides 16 and 8), the SepConv ASPP,
the decoder and the conv_upsample / pred_resize head.

TEST INFRASTRUCTURE ONLY.  Written from the reference's model files (cited per class), NOT from oracle/np_net.py: it shares
no builder, no padding helper, no resize and no loss with the oracle -- the graph is a tree of torch modules (nn.Conv2d with
groups / dilation, nn.BatchNorm2d, F.pad), the bilinear resize is a pair of dense interpolation matrices, the loss is
torch.log_softmax + gather, gradients come from torch autograd.  The only thing shared is the WEIGHTS, copied in by Keras layer
name.  What agrees between the two is therefore: layer order, names, channel widths, strides, atrous rates per output stride,
TensorFlow 'SAME' / ZeroPadding2D + 'VALID' padding, where the activations sit around a SepConv, BatchNorm epsilons, the ASPP
concat order, skip / shortcut wiring, squeeze-excite arithmetic, the resize convention, the loss reduction.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def tf_same_pad(size, k, stride, rate):
    """TensorFlow 'SAME' (tf.nn.convolution): out = ceil(size / stride); the total padding needed to place out windows of the
    dilated extent, the odd pixel at the END"""
    k_eff = (k - 1) * rate + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return total // 2, total - total // 2


class KConv(nn.Module):
    """one Keras Conv2D / DepthwiseConv2D: `pad` is 'same', 'valid' or an explicit ((top, bottom), (left, right)) that a
    ZeroPadding2D in front of a 'valid' conv applies"""

    def __init__(self, name, cin, cout, k, stride=1, rate=1, pad='same', bias=False, depthwise=False):
        super().__init__()
        self.kname, self.k, self.stride, self.rate, self.pad, self.depthwise = name, k, stride, rate, pad, depthwise
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, dilation=rate, groups=cin if depthwise else 1, bias=bias, padding=0)

    def forward(self, x):
        if self.pad == 'same':
            (t, b), (l, r) = tf_same_pad(x.shape[2], self.k, self.stride, self.rate), tf_same_pad(x.shape[3], self.k, self.stride, self.rate)
        elif self.pad == 'valid':
            t = b = l = r = 0
        else:
            (t, b), (l, r) = self.pad
        if t or b or l or r:
            x = F.pad(x, (l, r, t, b))
        return self.conv(x)


class KBN(nn.Module):
    def __init__(self, name, c, eps, momentum):
        super().__init__()
        self.kname = name
        # (Keras momentum m keeps m of the old value; torch's keeps 1 - momentum)
        self.bn = nn.BatchNorm2d(c, eps=eps, momentum=1.0 - momentum)

    def forward(self, x):
        return self.bn(x)


def hard_sigmoid(x):
    return F.relu6(x + 3.0) * (1.0 / 6.0)


def hard_swish(x):
    return hard_sigmoid(x) * x


# TensorFlow's resize kernel does its coordinate arithmetic in float32.  With FLOAT32_COORDS the matrices below round the way that
# kernel does (np.float32 scalars); without, they hold the real-valued weights -- the tests run the inference comparison on the
# real-valued ones (the difference, 2^-24 of the input extent per weight, is then visible and bounded) and the gradient comparison on the float32 ones
# (so that what remains is the graphs' and the differentiation's agreement, to 1e-9)
FLOAT32_COORDS = False


def resize_matrix(n_in, n_out, dtype=torch.float64):
    """tf.image.resize(method='bilinear') of TF 2 (half-pixel centres, no antialias): output pixel o samples the input at
    (o + 0.5) * n_in / n_out - 0.5, between the neighbours floor and ceil clamped to the image"""
    R = torch.zeros(n_out, n_in, dtype=dtype)
    f = np.float32 if FLOAT32_COORDS else float
    scale = f(n_in) / f(n_out)
    for o in range(n_out):
        s = (f(o) + f(0.5)) * scale - f(0.5)
        lo = math.floor(s)
        w = float(s - f(lo))
        a, b = min(max(lo, 0), n_in - 1), min(max(lo + 1, 0), n_in - 1)
        R[o, a] += 1.0 - w
        R[o, b] += w
    return R


def bilinear(x, H, W):
    if x.shape[2] == H and x.shape[3] == W:
        return x
    Rh, Rw = resize_matrix(x.shape[2], H, x.dtype), resize_matrix(x.shape[3], W, x.dtype)
    return torch.einsum('oh,nchw,pw->ncop', Rh, x, Rw)


class SepConvBN(nn.Module):
    """SepConv_BN, /root/reference/deeplabv3p/models/layers.py:74-111"""

    def __init__(self, prefix, cin, filters, stride=1, k=3, rate=1, depth_activation=False, eps=1e-3):
        super().__init__()
        if stride == 1:
            pad = 'same'
        else:
            k_eff = k + (k - 1) * (rate - 1)
            beg = (k_eff - 1) // 2
            pad = ((beg, k_eff - 1 - beg), (beg, k_eff - 1 - beg))
        self.depth_activation = depth_activation
        self.dw = KConv(prefix + '_depthwise', cin, cin, k, stride, rate, pad, depthwise=True)
        self.dw_bn = KBN(prefix + '_depthwise_BN', cin, eps, 0.99)
        self.pw = KConv(prefix + '_pointwise', cin, filters, 1)
        self.pw_bn = KBN(prefix + '_pointwise_BN', filters, eps, 0.99)

    def forward(self, x):
        if not self.depth_activation:
            x = F.relu(x)
        x = self.dw_bn(self.dw(x))
        if self.depth_activation:
            x = F.relu(x)
        x = self.pw_bn(self.pw(x))
        if self.depth_activation:
            x = F.relu(x)
        return x


class XceptionBlock(nn.Module):
    """_xception_block, /root/reference/deeplabv3p/models/deeplabv3p_xception.py:72-117 (shortcut conv: _conv2d_same :32-69 with
    kernel 1 -- at stride 2 a ZeroPadding2D((0, 0)) and 'valid')"""

    def __init__(self, prefix, cin, depths, skip_type, stride, rate=1, depth_activation=False):
        super().__init__()
        self.skip_type = skip_type
        c = cin
        seps = []
        for i in range(3):
            seps.append(SepConvBN(prefix + '_separable_conv%d' % (i + 1), c, depths[i], stride if i == 2 else 1, 3, rate, depth_activation))
            c = depths[i]
        self.seps = nn.ModuleList(seps)
        if skip_type == 'conv':
            self.shortcut = KConv(prefix + '_shortcut', cin, depths[-1], 1, stride, 1, 'same' if stride == 1 else 'valid')
            self.shortcut_bn = KBN(prefix + '_shortcut_BN', depths[-1], 1e-3, 0.99)

    def forward(self, x):
        r = x
        skip = None
        for i, s in enumerate(self.seps):
            r = s(r)
            if i == 1:
                skip = r
        if self.skip_type == 'conv':
            r = r + self.shortcut_bn(self.shortcut(x))
        elif self.skip_type == 'sum':
            r = r + x
        return r, skip


class XceptionBody(nn.Module):
    """Xception_body, /root/reference/deeplabv3p/models/deeplabv3p_xception.py:120-181"""
    out_channels, skip_channels = 2048, 256

    def __init__(self, OS):
        super().__init__()
        s16, r16, s32, r32 = {8: (1, 2, 1, 4), 16: (2, 1, 1, 2), 32: (2, 1, 2, 1)}[OS]
        self.conv1_1 = KConv('entry_flow_conv1_1', 3, 32, 3, 2)
        self.conv1_1_bn = KBN('entry_flow_conv1_1_BN', 32, 1e-3, 0.99)
        self.conv1_2 = KConv('entry_flow_conv1_2', 32, 64, 3, 1)
        self.conv1_2_bn = KBN('entry_flow_conv1_2_BN', 64, 1e-3, 0.99)
        self.b1 = XceptionBlock('entry_flow_block1', 64, [128] * 3, 'conv', 2)
        self.b2 = XceptionBlock('entry_flow_block2', 128, [256] * 3, 'conv', 2)
        self.b3 = XceptionBlock('entry_flow_block3', 256, [728] * 3, 'conv', s16)
        self.middle = nn.ModuleList([XceptionBlock('middle_flow_unit_%d' % (i + 1), 728, [728] * 3, 'sum', 1, r16) for i in range(16)])
        self.e1 = XceptionBlock('exit_flow_block1', 728, [728, 1024, 1024], 'conv', s32, r16)
        self.e2 = XceptionBlock('exit_flow_block2', 1024, [1536, 1536, 2048], 'none', 1, r32, depth_activation=True)

    def forward(self, x):
        x = F.relu(self.conv1_1_bn(self.conv1_1(x)))
        x = F.relu(self.conv1_2_bn(self.conv1_2(x)))
        x, _ = self.b1(x)
        x, skip = self.b2(x)
        x, _ = self.b3(x)
        for m in self.middle:
            x, _ = m(x)
        x, _ = self.e1(x)
        x, _ = self.e2(x)
        return x, skip


def _depth(v, divisor=8, min_value=None):
    """/root/reference/deeplabv3p/models/deeplabv3p_mobilenetv3.py:112-119"""
    min_value = divisor if min_value is None else min_value
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class SE(nn.Module):
    """_se_block, deeplabv3p_mobilenetv3.py:122-146: mean over the map -> 1x1 (+bias) -> ReLU -> 1x1 (+bias) -> hard sigmoid -> scale"""

    def __init__(self, prefix, c, se_ratio):
        super().__init__()
        self.reduce = KConv(prefix + 'squeeze_excite/Conv', c, _depth(c * se_ratio), 1, bias=True)
        self.expand = KConv(prefix + 'squeeze_excite/Conv_1', _depth(c * se_ratio), c, 1, bias=True)

    def forward(self, x):
        s = x.mean(dim=(2, 3), keepdim=True)
        s = hard_sigmoid(self.expand(F.relu(self.reduce(s))))
        return x * s


class InvertedResBlock(nn.Module):
    """_inverted_res_block, deeplabv3p_mobilenetv3.py:149-201 (block 0 has no expand conv; the depthwise conv is padding='same'
    at every stride)"""

    def __init__(self, cin, expansion, filters, k, stride, se_ratio, activation, block_id, skip_connection=False, rate=1):
        super().__init__()
        prefix = 'expanded_conv_%d/' % block_id if block_id else 'expanded_conv/'
        self.act, self.skip_connection = activation, skip_connection
        mid = _depth(cin * expansion)
        self.has_expand = bool(block_id)
        if self.has_expand:
            self.expand = KConv(prefix + 'expand', cin, mid, 1)
            self.expand_bn = KBN(prefix + 'expand/BatchNorm', mid, 1e-3, 0.999)
        self.dw = KConv(prefix + 'depthwise/Conv', mid, mid, k, stride, rate, 'same', depthwise=True)
        self.dw_bn = KBN(prefix + 'depthwise/BatchNorm', mid, 1e-3, 0.999)
        self.se = SE(prefix, mid, se_ratio) if se_ratio else None
        self.project = KConv(prefix + 'project', mid, filters, 1)
        self.project_bn = KBN(prefix + 'project/BatchNorm', filters, 1e-3, 0.999)
        self.out_channels = filters

    def forward(self, x):
        y = x
        if self.has_expand:
            y = self.act(self.expand_bn(self.expand(y)))
        y = self.act(self.dw_bn(self.dw(y)))
        if self.se is not None:
            y = self.se(y)
        y = self.project_bn(self.project(y))
        return x + y if self.skip_connection else y


class MobileNetV3Body(nn.Module):
    """MobileNetV3 / MobileNetV3Large / MobileNetV3Small with include_top=False as the DeepLab constructors use them,
    deeplabv3p_mobilenetv3.py:204-432 (stem: ZeroPadding2D(correct_pad) + 3x3 stride 2 'valid', hard swish), :436-515 (small),
    :518-607 (large); the feature handed on is the last block's output (`final_feature`, :358), not Conv_1"""

    def __init__(self, OS, kind='large', input_hw=(None, None)):
        super().__init__()
        s16, r16, s32, r32 = {8: (1, 2, 1, 4), 16: (2, 1, 1, 2), 32: (2, 1, 2, 1)}[OS]
        H, W = input_hw
        # correct_pad (:50-72): kernel // 2 on both sides, one less in front when the size is even
        pad = ((1 - (1 - H % 2), 1), (1 - (1 - W % 2), 1))
        self.stem = KConv('Conv', 3, 16, 3, 2, 1, pad)
        self.stem_bn = KBN('Conv/BatchNorm', 16, 1e-3, 0.999)
        hs, re, se = hard_swish, F.relu, 0.25
        if kind == 'large':
            #        expansion filters k stride se activation skip rate
            rows = [(1, 16, 3, 1, None, re, True, 1), (4, 24, 3, 2, None, re, False, 1), (3, 24, 3, 1, None, re, True, 1),
                    (3, 40, 5, 2, se, re, False, 1), (3, 40, 5, 1, se, re, True, 1), (3, 40, 5, 1, se, re, True, 1),
                    (6, 80, 3, s16, None, hs, False, 1), (2.5, 80, 3, 1, None, hs, True, r16), (2.3, 80, 3, 1, None, hs, True, r16),
                    (2.3, 80, 3, 1, None, hs, True, r16), (6, 112, 3, 1, se, hs, False, r16), (6, 112, 3, 1, se, hs, True, r16),
                    (6, 160, 5, s32, se, hs, False, r16), (6, 160, 5, 1, se, hs, True, r32), (6, 160, 5, 1, se, hs, True, r32)]
            self.skip_after = 2
        else:
            rows = [(1, 16, 3, 2, se, re, False, 1), (72. / 16, 24, 3, 2, None, re, False, 1), (88. / 24, 24, 3, 1, None, re, True, 1),
                    (4, 40, 5, s16, se, hs, False, 1), (6, 40, 5, 1, se, hs, True, r16), (6, 40, 5, 1, se, hs, True, r16),
                    (3, 48, 5, 1, se, hs, False, r16), (3, 48, 5, 1, se, hs, True, r16), (6, 96, 5, s32, se, hs, False, r16),
                    (6, 96, 5, 1, se, hs, True, r32), (6, 96, 5, 1, se, hs, True, r32)]
            self.skip_after = 0
        blocks, c = [], 16
        for i, (e, f, k, s, ser, a, sk, r) in enumerate(rows):
            blocks.append(InvertedResBlock(c, e, _depth(f), k, s, ser, a, i, sk, r))
            c = _depth(f)
        self.blocks = nn.ModuleList(blocks)
        self.out_channels = c
        self.skip_channels = _depth(rows[self.skip_after][1])

    def forward(self, x):
        x = hard_swish(self.stem_bn(self.stem(x)))
        skip = None
        for i, b in enumerate(self.blocks):
            x = b(x)
            if i == self.skip_after:
                skip = x
        return x, skip


class MNV2Block(nn.Module):
    """_inverted_res_block, /root/reference/deeplabv3p/models/deeplabv3p_mobilenetv2.py:38-74: 1x1 expand to expansion * cin (not
    rounded; none in block 0) -> 3x3 depthwise 'same' at (stride, rate) -> 1x1 project to make_divisible(filters), BatchNorm(1e-3,
    0.999) behind each, ReLU6 behind the first two"""

    def __init__(self, cin, expansion, stride, filters, block_id, skip_connection, rate=1):
        super().__init__()
        prefix = 'expanded_conv_%d_' % block_id if block_id else 'expanded_conv_'
        mid = expansion * cin
        self.has_expand = bool(block_id)
        if self.has_expand:
            self.expand = KConv(prefix + 'expand', cin, mid, 1)
            self.expand_bn = KBN(prefix + 'expand_BN', mid, 1e-3, 0.999)
        self.dw = KConv(prefix + 'depthwise', mid, mid, 3, stride, rate, 'same', depthwise=True)
        self.dw_bn = KBN(prefix + 'depthwise_BN', mid, 1e-3, 0.999)
        self.out_channels = _depth(filters)
        self.project = KConv(prefix + 'project', mid, self.out_channels, 1)
        self.project_bn = KBN(prefix + 'project_BN', self.out_channels, 1e-3, 0.999)
        self.skip_connection = skip_connection

    def forward(self, x):
        y = x
        if self.has_expand:
            y = F.relu6(self.expand_bn(self.expand(y)))
        y = F.relu6(self.dw_bn(self.dw(y)))
        y = self.project_bn(self.project(y))
        return x + y if self.skip_connection else y


class MobileNetV2Body(nn.Module):
    """MobileNetV2_body, deeplabv3p_mobilenetv2.py:77-199 (alpha = 1): 3x3 stride 2 'same' stem + 17 inverted residual blocks with the
    output-stride table; the feature handed on is block 16's output (Conv_1 is not part of the DeepLab graph), the skip block 2's"""
    out_channels, skip_channels = 320, 24

    def __init__(self, OS):
        super().__init__()
        s16, r16, s32, r32 = {8: (1, 2, 1, 4), 16: (2, 1, 1, 2), 32: (2, 1, 2, 1)}[OS]
        self.stem = KConv('Conv', 3, 32, 3, 2)
        self.stem_bn = KBN('Conv_BN', 32, 1e-3, 0.999)
        #        filters stride expansion skip rate
        rows = [(16, 1, 1, False, 1), (24, 2, 6, False, 1), (24, 1, 6, True, 1), (32, 2, 6, False, 1), (32, 1, 6, True, 1), (32, 1, 6, True, 1),
                (64, s16, 6, False, 1), (64, 1, 6, True, r16), (64, 1, 6, True, r16), (64, 1, 6, True, r16),
                (96, 1, 6, False, r16), (96, 1, 6, True, r16), (96, 1, 6, True, r16),
                (160, s32, 6, False, r16), (160, 1, 6, True, r32), (160, 1, 6, True, r32), (320, 1, 6, False, r32)]
        blocks, c = [], 32
        for i, (f, st, e, sk, r) in enumerate(rows):
            blocks.append(MNV2Block(c, e, st, f, i, sk, r))
            c = blocks[-1].out_channels
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        x = F.relu6(self.stem_bn(self.stem(x)))
        skip = None
        for i, b in enumerate(self.blocks):
            x = b(x)
            if i == 2:
                skip = x
        return x, skip


class ResBlock(nn.Module):
    """identity_block / conv_block, /root/reference/deeplabv3p/models/deeplabv3p_resnet50.py:32-75, :78-139: 1x1 (the stride of a
    conv_block sits HERE) -> 3x3 'same' at the block's atrous rate -> 1x1, every conv with a bias, BatchNorm(1e-3, 0.99) behind each;
    the conv_block's shortcut is a strided 1x1 + BatchNorm"""

    def __init__(self, cin, filters, stage, block, conv_shortcut, stride=1, rate=1):
        super().__init__()
        f1, f2, f3 = filters
        base, bn = 'res%d%s_branch' % (stage, block), 'bn%d%s_branch' % (stage, block)
        self.a = KConv(base + '2a', cin, f1, 1, stride, 1, 'valid', bias=True)
        self.a_bn = KBN(bn + '2a', f1, 1e-3, 0.99)
        self.b = KConv(base + '2b', f1, f2, 3, 1, rate, 'same', bias=True)
        self.b_bn = KBN(bn + '2b', f2, 1e-3, 0.99)
        self.c = KConv(base + '2c', f2, f3, 1, 1, 1, 'valid', bias=True)
        self.c_bn = KBN(bn + '2c', f3, 1e-3, 0.99)
        self.conv_shortcut = conv_shortcut
        if conv_shortcut:
            self.s = KConv(base + '1', cin, f3, 1, stride, 1, 'valid', bias=True)
            self.s_bn = KBN(bn + '1', f3, 1e-3, 0.99)

    def forward(self, x):
        y = F.relu(self.a_bn(self.a(x)))
        y = F.relu(self.b_bn(self.b(y)))
        y = self.c_bn(self.c(y))
        # (Keras builds the main path's three convs first, then the shortcut conv: the order the names are recorded in)
        sc = self.s_bn(self.s(x)) if self.conv_shortcut else x
        return F.relu(y + sc)


class ResNet50Body(nn.Module):
    """ResNet50 with include_top=False as Deeplabv3pResNet50 uses it, deeplabv3p_resnet50.py:142-292: conv1_pad (3, 3) + 7x7 stride 2
    'valid' (+ bias) -> BatchNorm -> ReLU -> pool1_pad (1, 1) + 3x3 max pooling stride 2; stages 2-5 with the output-stride table
    (:158-177): the stride of stage 4 / 5 and the atrous rates of their blocks; the skip feature is the output of stage 2"""
    out_channels, skip_channels = 2048, 256

    def __init__(self, OS):
        super().__init__()
        s16, r16, s32, r32 = {8: (1, 2, 1, 4), 16: (2, 1, 1, 2), 32: (2, 1, 2, 1)}[OS]
        self.conv1 = KConv('conv1', 3, 64, 7, 2, 1, ((3, 3), (3, 3)), bias=True)
        self.bn1 = KBN('bn_conv1', 64, 1e-3, 0.99)
        spec = [(2, 'a', [64, 64, 256], True, 1, 1), (2, 'b', [64, 64, 256], False, 1, 1), (2, 'c', [64, 64, 256], False, 1, 1),
                (3, 'a', [128, 128, 512], True, 2, 1)] + [(3, b, [128, 128, 512], False, 1, 1) for b in 'bcd'] + \
               [(4, 'a', [256, 256, 1024], True, s16, 1)] + [(4, b, [256, 256, 1024], False, 1, r16) for b in 'bcdef'] + \
               [(5, 'a', [512, 512, 2048], True, s32, r16), (5, 'b', [512, 512, 2048], False, 1, r32), (5, 'c', [512, 512, 2048], False, 1, r32)]
        blocks, c = [], 64
        for stage, block, filters, conv_sc, stride, rate in spec:
            blocks.append(ResBlock(c, filters, stage, block, conv_sc, stride, rate))
            c = filters[2]
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(F.pad(x, (1, 1, 1, 1)), 3, 2)          # ZeroPadding2D then 'valid' pooling: the zeros take part (inputs are >= 0)
        skip = None
        for i, b in enumerate(self.blocks):
            x = b(x)
            if i == 2:
                skip = x
        return x, skip


class ASPP(nn.Module):
    """ASPP_block, /root/reference/deeplabv3p/models/layers.py:114-163: concat order [image pooling, 1x1, rate a, rate b, rate c]"""

    def __init__(self, cin, OS):
        super().__init__()
        rates = {8: (12, 24, 36), 16: (6, 12, 18), 32: (3, 6, 9)}[OS]
        self.pool = KConv('image_pooling', cin, 256, 1)
        self.pool_bn = KBN('image_pooling_BN', 256, 1e-5, 0.99)
        self.b0 = KConv('aspp0', cin, 256, 1)
        self.b0_bn = KBN('aspp0_BN', 256, 1e-5, 0.99)
        self.branches = nn.ModuleList([SepConvBN('aspp%d' % (i + 1), cin, 256, rate=r, depth_activation=True, eps=1e-5)
                                       for i, r in enumerate(rates)])
        self.proj = KConv('concat_projection', 256 * 5, 256, 1)
        self.proj_bn = KBN('concat_projection_BN', 256, 1e-5, 0.99)

    def forward(self, x, dropout_mask=None):
        H, W = x.shape[2:]
        b4 = F.avg_pool2d(x, (H, W))
        b4 = bilinear(F.relu(self.pool_bn(self.pool(b4))), H, W)
        b0 = F.relu(self.b0_bn(self.b0(x)))
        x = torch.cat([b4, b0] + [b(x) for b in self.branches], dim=1)
        x = F.relu(self.proj_bn(self.proj(x)))
        if self.training and dropout_mask is not None:      # Dropout(0.5): kept entries scaled by 1 / (1 - rate)
            x = x * dropout_mask * 2.0
        return x


class ASPPLite(nn.Module):
    """ASPP_Lite_block, layers.py:166-196: the image-pooling branch and the 1x1 branch only, concat [pooling, 1x1]"""

    def __init__(self, cin):
        super().__init__()
        self.pool = KConv('image_pooling', cin, 256, 1)
        self.pool_bn = KBN('image_pooling_BN', 256, 1e-5, 0.99)
        self.b0 = KConv('aspp0', cin, 256, 1)
        self.b0_bn = KBN('aspp0_BN', 256, 1e-5, 0.99)
        self.proj = KConv('concat_projection', 512, 256, 1)
        self.proj_bn = KBN('concat_projection_BN', 256, 1e-5, 0.99)

    def forward(self, x, dropout_mask=None):
        H, W = x.shape[2:]
        b4 = bilinear(F.relu(self.pool_bn(self.pool(F.avg_pool2d(x, (H, W))))), H, W)
        b0 = F.relu(self.b0_bn(self.b0(x)))
        x = F.relu(self.proj_bn(self.proj(torch.cat([b4, b0], dim=1))))
        if self.training and dropout_mask is not None:
            x = x * dropout_mask * 2.0
        return x


class Decoder(nn.Module):
    """Decoder_block, layers.py:199-219"""

    def __init__(self, skip_channels):
        super().__init__()
        self.fp = KConv('feature_projection0', skip_channels, 48, 1)
        self.fp_bn = KBN('feature_projection0_BN', 48, 1e-5, 0.99)
        self.c0 = SepConvBN('decoder_conv0', 256 + 48, 256, depth_activation=True, eps=1e-5)
        self.c1 = SepConvBN('decoder_conv1', 256, 256, depth_activation=True, eps=1e-5)

    def forward(self, x, skip):
        x = bilinear(x, skip.shape[2], skip.shape[3])
        s = F.relu(self.fp_bn(self.fp(skip)))
        return self.c1(self.c0(torch.cat([x, s], dim=1)))


class DeepLabV3Plus(nn.Module):
    """get_deeplabv3p_model, /root/reference/deeplabv3p/model.py:51-83 over Deeplabv3pXception (deeplabv3p_xception.py:184-240) /
    Deeplabv3pMobileNetV3Large (deeplabv3p_mobilenetv3.py:615-681) / ...Small (:754-820): body -> ASPP -> decoder -> [the 21-class
    logits_semantic stub cut at layers[-5]] conv_upsample (1x1 + bias) -> pred_resize to the input size"""

    def __init__(self, model_type, num_classes, input_hw, OS):
        super().__init__()
        if model_type == 'xception':
            self.body = XceptionBody(OS)
        elif model_type == 'resnet50':
            self.body = ResNet50Body(OS)
        elif model_type.startswith('mobilenetv2'):
            self.body = MobileNetV2Body(OS)
        else:
            self.body = MobileNetV3Body(OS, 'large' if 'large' in model_type else 'small', input_hw)
        # Deeplabv3pLite* (deeplabv3p_mobilenetv2.py:273-351, deeplabv3p_mobilenetv3.py:684-751): ASPP-Lite, no decoder
        self.lite = model_type.endswith('_lite')
        if self.lite:
            self.aspp = ASPPLite(self.body.out_channels)
        else:
            self.aspp = ASPP(self.body.out_channels, OS)
            self.decoder = Decoder(self.body.skip_channels)
        self.head = KConv('conv_upsample', 256, num_classes, 1, bias=True)
        self.input_hw = input_hw

    def forward(self, x, dropout_mask=None):
        f, skip = self.body(x)
        y = self.aspp(f, dropout_mask)
        if not self.lite:
            y = self.decoder(y, skip)
        return bilinear(self.head(y), *self.input_hw)

    # ---- weights by Keras layer name; conv outputs by Keras layer name ----
    def load_keras(self, params):
        """params: {'<layer>/kernel' (HWIO) | '/depthwise_kernel' (HWC1) | '/bias' | '/gamma' | '/beta' | '/moving_mean' |
        '/moving_variance'} -> this tree.  Every entry of `params` must be consumed and every module fed."""
        used = set()

        def take(name):
            used.add(name)
            return torch.from_numpy(np.ascontiguousarray(params[name])).double()
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, KConv):
                    if m.depthwise:
                        m.conv.weight.copy_(take(m.kname + '/depthwise_kernel').permute(2, 3, 0, 1))
                    else:
                        m.conv.weight.copy_(take(m.kname + '/kernel').permute(3, 2, 0, 1))
                    if m.conv.bias is not None:
                        m.conv.bias.copy_(take(m.kname + '/bias'))
                elif isinstance(m, KBN):
                    m.bn.weight.copy_(take(m.kname + '/gamma'))
                    m.bn.bias.copy_(take(m.kname + '/beta'))
                    m.bn.running_mean.copy_(take(m.kname + '/moving_mean'))
                    m.bn.running_var.copy_(take(m.kname + '/moving_variance'))
        assert used == set(params), sorted(set(params) - used)[:5]

    def keras_grads(self):
        out = {}
        for m in self.modules():
            if isinstance(m, KConv):
                g = m.conv.weight.grad
                if m.depthwise:
                    out[m.kname + '/depthwise_kernel'] = g.permute(2, 3, 0, 1).numpy()
                else:
                    out[m.kname + '/kernel'] = g.permute(2, 3, 1, 0).numpy()
                if m.conv.bias is not None:
                    out[m.kname + '/bias'] = m.conv.bias.grad.numpy()
            elif isinstance(m, KBN):
                out[m.kname + '/gamma'] = m.bn.weight.grad.numpy()
                out[m.kname + '/beta'] = m.bn.bias.grad.numpy()
        return out

    def record_convs(self):
        """forward hooks: {Keras layer name: conv output, NHWC numpy}"""
        rec = {}
        hooks = [m.register_forward_hook(lambda mod, a, out: rec.__setitem__(mod.kname, out.detach().permute(0, 2, 3, 1).numpy()))
                 for m in self.modules() if isinstance(m, KConv)]
        return rec, hooks


def keras_sparse_ce(logits, labels, ignore_index=255):
    """the training loss (/root/reference/deeplabv3p/loss.py sparse CE with ignore_index over Keras' categorical_crossentropy on
    probabilities): pixels labelled `ignore_index` contribute nothing, the probability is clipped to [1e-7, 1 - 1e-7], and the
    mean runs over ALL pixels.  logits (N, C, H, W); labels (N, H, W) integers"""
    N, C, H, W = logits.shape
    p = torch.softmax(logits, dim=1)
    lab = labels.long()
    keep = lab != ignore_index
    pt = p.gather(1, lab.clamp(0, C - 1).unsqueeze(1)).squeeze(1)
    pt = pt.clamp(1e-7, 1.0 - 1e-7)
    return -(torch.log(pt) * keep).sum() / (N * H * W)
