"""DeepLabV3+ block library on the static graph: the counterparts of the reference's
deeplabv3p/models/layers.py (SepConv_BN :74-111, ASPP_block :114-163, ASPP_Lite_block :166-196,
Decoder_block :199-219), same names, argument meaning and layer names -- but every block lowers to
fused HIP ops instead of wrapping tf.keras layers."""
from .graph import ACT_NONE, ACT_RELU, ACT_RELU6, Value  # noqa: F401


def SepConv_BN(g, x, filters, prefix, stride=1, kernel_size=3, rate=1, depth_activation=False, epsilon=1e-3,
               out=None, out_group=None, out_goff=0):
    """SepConv with BN between depthwise & pointwise (reference layers.py:74-111).
    `out`/`out_group`: write the pointwise output straight into a Concatenate slice."""
    if stride == 1:
        depth_padding = 'same'
    else:
        kernel_size_effective = kernel_size + (kernel_size - 1) * (rate - 1)
        pad_total = kernel_size_effective - 1
        pad_beg = pad_total // 2
        pad_end = pad_total - pad_beg
        depth_padding = (pad_beg, pad_end, pad_beg, pad_end)   # ZeroPadding2D + 'valid'
    if not depth_activation:
        x = g.relu(x)
    x = g.dwconv2d(x, kernel_size, prefix + '_depthwise', stride=stride, rate=rate, padding=depth_padding)
    x = g.batchnorm(x, prefix + '_depthwise_BN', eps=epsilon)
    if depth_activation:
        x = g.relu(x)
    x = g.conv2d(x, filters, 1, prefix + '_pointwise', out=out)
    x = g.batchnorm(x, prefix + '_pointwise_BN', eps=epsilon, group=out_group, goff=out_goff)
    if depth_activation:
        x = g.relu(x)
    return x


def _image_pooling_branch(g, x, out, group, goff):
    """reference layers.py:131-138: AveragePooling2D(full map) -> 1x1(256) -> BN(1e-5) -> ReLU -> resize"""
    H, W, _ = x.shape
    b4 = g.global_avgpool(x)
    b4 = g.conv2d(b4, 256, 1, 'image_pooling')
    b4 = g.batchnorm(b4, 'image_pooling_BN', eps=1e-5, group=group, goff=goff)
    b4 = g.relu(b4)
    return g.resize(b4, H, W, 'aspp_resize', out=out)


def ASPP_block(g, x, OS):
    """branching for Atrous Spatial Pyramid Pooling (reference layers.py:114-163)"""
    if OS == 8:
        atrous_rates = (12, 24, 36)
    elif OS == 16:
        atrous_rates = (6, 12, 18)
    elif OS == 32:
        atrous_rates = (3, 6, 9)
    else:
        raise ValueError('invalid output stride', OS)
    H, W, _ = x.shape
    x_in = x
    # Concatenate([b4, b0, b1, b2, b3]) (layers.py:155): the five branches write their raw outputs
    # into channel slices of one buffer; their BNs own slices of one coefficient group
    base, slices, group = g.concat_buffer(H, W, [256] * 5, 'aspp_concat')
    b4 = _image_pooling_branch(g, x, slices[0][0], group, slices[0][1])
    b0 = g.conv2d(x, 256, 1, 'aspp0', out=slices[1][0])
    b0 = g.batchnorm(b0, 'aspp0_BN', eps=1e-5, group=group, goff=slices[1][1])
    branches = [b4, g.relu(b0, 'aspp0_activation')]
    for i, r in enumerate(atrous_rates):
        branches.append(SepConv_BN(g, x, 256, 'aspp%d' % (i + 1), rate=r, depth_activation=True, epsilon=1e-5,
                                   out=slices[2 + i][0], out_group=group, out_goff=slices[2 + i][1]))
    _atrous_first(g, x_in)
    x = g.concat_value(base, group, ACT_RELU, branches)          # Concatenate()([b4, b0, b1, b2, b3]) (:155)
    x = g.conv2d(x, 256, 1, 'concat_projection')
    x = g.batchnorm(x, 'concat_projection_BN', eps=1e-5)
    x = g.relu(x)
    return g.dropout(x, 0.5)


def _atrous_first(g, x_in):
    """Execution order only (layers and weights keep the reference's order): run the three atrous
    depthwise convs directly after the op that produced the ASPP input, while that tensor is still
    resident in the Infinity Cache, then the pooling / 1x1 branches.  Rate 6, 12, then 18 (measured:
    the rate-18 kernel takes 9.6 us behind the two other readers of x, 11.1 us straight behind the
    projection GEMM whose 22 MB of output is still being written back)."""
    import os
    order = int(os.environ.get('DL3P_ASPP_ORDER', '2'))
    if order == 3:
        return
    names = ['aspp3_depthwise', 'aspp2_depthwise', 'aspp1_depthwise']
    if order in (2, 4):
        names = names[::-1]
    moved = []
    for n in names:
        for op in g.ops:
            if getattr(op, 'name', None) in (n, n + '_BN') and op.kind in ('conv_dw', 'bn'):
                moved.append(op)
    first = min(i for i, op in enumerate(g.ops) if getattr(op, 'x', None) is not None and op.x.tensor is x_in.tensor
                and op.kind in ('gap', 'conv_pw', 'conv_dw'))
    rest = [op for op in g.ops if op not in moved]
    if order in (1, 4):      # pooling branch (gap, 1x1 conv, BN, broadcast) first, then the atrous convs
        pool = [op for op in rest[first:] if getattr(op, 'name', '') in ('image_pooling', 'image_pooling_BN', 'aspp_resize')
                or op.kind == 'gap']
        rest2 = [op for op in rest if op not in pool]
        first2 = first
        g.ops[:] = rest2[:first2] + pool + moved + rest2[first2:]
        return
    g.ops[:] = rest[:first] + moved + rest[first:]


def ASPP_Lite_block(g, x):
    """global pooling + 1x1 branch only (reference layers.py:166-196)"""
    H, W, _ = x.shape
    base, slices, group = g.concat_buffer(H, W, [256] * 2, 'aspp_concat')
    b4 = _image_pooling_branch(g, x, slices[0][0], group, slices[0][1])
    b0 = g.conv2d(x, 256, 1, 'aspp0', out=slices[1][0])
    b0 = g.batchnorm(b0, 'aspp0_BN', eps=1e-5, group=group, goff=slices[1][1])
    b0 = g.relu(b0, 'aspp0_activation')
    x = g.concat_value(base, group, ACT_RELU, [b4, b0])          # Concatenate()([b4, b0]) (:189)
    x = g.conv2d(x, 256, 1, 'concat_projection')
    x = g.batchnorm(x, 'concat_projection_BN', eps=1e-5)
    x = g.relu(x)
    return g.dropout(x, 0.5)


def Decoder_block(g, x, skip_feature):
    """DeepLab v3+ decoder (reference layers.py:199-219): bilinear-resize + concat + 2 SepConv"""
    H, W, _ = skip_feature.shape
    base, slices, group = g.concat_buffer(H, W, [x.shape[2], 48], 'decoder_concat')
    # the resized ASPP output is already activated (>= 0): identity coefficients + ReLU is a no-op on it
    xr = g.resize(x, H, W, 'decoder_resize', out=slices[0][0])
    s = g.conv2d(skip_feature, 48, 1, 'feature_projection0', out=slices[1][0])
    s = g.batchnorm(s, 'feature_projection0_BN', eps=1e-5, group=group, goff=slices[1][1])
    s = g.relu(s)
    x = g.concat_value(base, group, ACT_RELU, [xr, s])           # Concatenate()([x, skip_feature]) (:214)
    x = SepConv_BN(g, x, 256, 'decoder_conv0', depth_activation=True, epsilon=1e-5)
    x = SepConv_BN(g, x, 256, 'decoder_conv1', depth_activation=True, epsilon=1e-5)
    return x
