"""Static layer graph of a DeepLabV3+ model (pure Python, no device needed).

The reference builds its models with the Keras functional API (deeplabv3p/models/*.py); here the same
calls record a static graph: Keras-like `Layer` records (names, weights, `trainable`) for the user-facing
façade, and fused execution ops for the HIP executor.  A conv output is stored RAW; the
BatchNormalization + activation that follow it are carried as a lazy per-channel affine on the `Value`
and applied inside the consuming kernel's prologue (include/dl3p.h), so normalised activations are
never written to HBM.  Concatenate is a set of channel-slice views of one buffer.
"""
import math
import numpy as np

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4
# device-side channel padding granule: 4 floats (16 bytes) on the fp32 path; the bf16 GEMMs read 8-element (16-byte)
# chunks, so a mixed-precision graph pads the class dimension and the im2col rows to multiples of 8 (mixed_precision.py)
CHANNEL_ALIGN = 4
L2_FACTOR = 2e-5  # reference deeplabv3p/models/layers.py:12


def same_pad(in_size, k, stride, rate):
    k_eff = k + (k - 1) * (rate - 1)
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return out, total // 2


def conv_geometry(H, W, k, stride, rate, padding):
    """'same' | 'valid' | (pt,pb,pl,pr) explicit ZeroPadding2D + VALID  ->  Ho, Wo, pad_t, pad_l"""
    if padding == 'same':
        Ho, pt = same_pad(H, k, stride, rate)
        Wo, pl = same_pad(W, k, stride, rate)
        return Ho, Wo, pt, pl
    if padding == 'valid':
        padding = (0, 0, 0, 0)
    pt, pb, pl, pr = padding
    k_eff = k + (k - 1) * (rate - 1)
    return (H + pt + pb - k_eff) // stride + 1, (W + pl + pr - k_eff) // stride + 1, pt, pl


class Param:
    """one Keras weight.  `shape` is the Keras shape; `dev_shape` the (possibly channel-padded)
    device shape (the class logits are padded to a multiple of 4 columns)."""

    def __init__(self, layer, key, shape, init, l2=0.0, trainable=True, dev_shape=None):
        self.layer, self.key, self.shape, self.init, self.l2 = layer, key, tuple(shape), init, l2
        self.weight_trainable = trainable   # False for BN moving statistics
        self.dev_shape = tuple(dev_shape) if dev_shape is not None else self.shape
        self.value = None                   # host copy (numpy float32), Keras shape

    @property
    def name(self):
        return self.layer.name + '/' + self.key

    @property
    def trainable(self):
        return self.weight_trainable and self.layer.trainable

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def dev_size(self):
        return int(np.prod(self.dev_shape))


class Layer:
    """Keras-like layer record: model.layers[i].name / .trainable / .get_weights()"""

    def __init__(self, name, kind, output_shape=None):
        self.name, self.kind, self.output_shape = name, kind, output_shape
        self.trainable = True
        self.params = []
        self.inbound = []       # the Keras layers whose outputs this layer is called on (functional-API edges)

    def add_param(self, key, shape, init, **kw):
        p = Param(self, key, shape, init, **kw)
        self.params.append(p)
        return p

    def count_params(self):
        return sum(p.size for p in self.params)

    def get_weights(self):
        return [p.value for p in self.params]

    def set_weights(self, ws):
        assert len(ws) == len(self.params)
        for p, w in zip(self.params, ws):
            w = np.asarray(w, dtype=np.float32)
            assert w.shape == p.shape, (p.name, w.shape, p.shape)
            p.value = w.copy()

    def __repr__(self):
        return '<%s %s>' % (self.kind, self.name)


class Tensor:
    """a device buffer of shape (N, H, W, C) or a channel slice [c0, c0+C) of `base`"""
    _next = 0

    def __init__(self, H, W, C, name, base=None, c0=0):
        self.H, self.W, self.C, self.name, self.base, self.c0 = H, W, C, name, base, c0
        self.id = Tensor._next
        Tensor._next += 1
        self.requires_grad = False

    @property
    def root(self):
        return self.base if self.base is not None else self

    @property
    def ld(self):
        return self.root.C

    def slice(self, c0, C, name=None):
        assert self.base is None and c0 + C <= self.C
        return Tensor(self.H, self.W, C, name or '%s[%d:%d]' % (self.name, c0, c0 + C), base=self, c0=c0)


class CoefGroup:
    """contiguous per-channel (scale, shift) arrays read by a consumer's prologue; a concat value
    concatenates the slots of its branches"""
    _next = 0

    def __init__(self, C):
        self.C = C
        self.id = CoefGroup._next
        CoefGroup._next += 1
        self.identity = [True] * C    # channels not owned by a BN keep (1, 0)


class BNSpec:
    def __init__(self, layer, C, eps, momentum, group, offset):
        self.layer, self.C, self.eps, self.momentum, self.group, self.offset = layer, C, eps, momentum, group, offset
        self.act = ACT_NONE      # activation applied on top of this BN (one per BN in these graphs)
        self.z = None            # Tensor it normalises

    @property
    def name(self):
        return self.layer.name


class Value:
    """act(tensor * scale + shift): what a Keras tensor is, lazily"""

    def __init__(self, tensor, group=None, goff=0, act=ACT_NONE, bn=None, klayer=None):
        self.tensor, self.group, self.goff, self.act, self.bn = tensor, group, goff, act, bn
        self.klayer = klayer      # the Keras layer this tensor is the output of

    def from_layer(self, layer):
        """the same lazy value as the output tensor of another Keras layer (a copy: `self` may have other consumers)"""
        import copy
        v = copy.copy(self)
        v.klayer = layer
        return v

    @property
    def shape(self):
        return (self.tensor.H, self.tensor.W, self.tensor.C)

    @property
    def is_plain(self):
        return self.group is None and self.act == ACT_NONE


class Op:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return '<Op %s %s>' % (self.kind, getattr(self, 'name', ''))


def glorot_uniform(rng, shape, fan_in, fan_out):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def he_normal(rng, shape, fan_in):
    """Keras `he_normal` = VarianceScaling(2, 'fan_in', 'truncated_normal'): N(0, s) cut at +-2 s with
    s = sqrt(2 / fan_in) / 0.87962566103423978 (so that the truncated law has variance 2 / fan_in)"""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


class GraphBuilder:
    """records layers / ops while the model functions (layers.py, mobilenetv2.py, ...) run"""

    def __init__(self, input_shape, name='model', seed=0):
        H, W, C = input_shape
        self.name = name
        self.align = CHANNEL_ALIGN
        self.rng = np.random.default_rng(seed)
        self.layers = []
        self.layer_by_name = {}
        self.ops = []
        self.bns = []
        self.tensors = []
        self.groups = []
        self.dropout_count = 0
        self.act_views = {}
        inp = self.new_tensor(H, W, C, 'image_input')
        self.input = Value(inp, klayer=self.add_layer('image_input', 'InputLayer', (H, W, C)))
        self.input_shape = (H, W, C)
        self.taps = {}

    # ---- bookkeeping ------------------------------------------------------------------
    def add_layer(self, name, kind, output_shape=None, inbound=()):
        if name is None:
            base = {'ReLU': 're_lu', 'Add': 'add', 'Concatenate': 'concatenate', 'Dropout': 'dropout',
                    'ZeroPadding2D': 'zero_padding2d', 'AveragePooling2D': 'average_pooling2d',
                    'Activation': 'activation', 'Multiply': 'multiply', 'Reshape': 'reshape',
                    'GlobalAveragePooling2D': 'global_average_pooling2d', 'MaxPooling2D': 'max_pooling2d'}.get(kind, kind.lower())
            n = sum(1 for l in self.layers if l.kind == kind and l.name.startswith(base))
            name = base if n == 0 else '%s_%d' % (base, n)
        assert name not in self.layer_by_name, 'duplicate layer name ' + name
        layer = Layer(name, kind, output_shape)
        for v in inbound:
            src = v.klayer if isinstance(v, Value) else v
            assert src is not None, 'layer %s: an input tensor has no producing Keras layer' % name
            layer.inbound.append(src)
        self.layers.append(layer)
        self.layer_by_name[name] = layer
        return layer

    def new_tensor(self, H, W, C, name):
        t = Tensor(H, W, C, name)
        self.tensors.append(t)
        return t

    def new_group(self, C):
        g = CoefGroup(C)
        self.groups.append(g)
        return g

    def tap(self, name, value):
        self.taps[name] = value
        return value

    # ---- Keras layers -----------------------------------------------------------------
    def conv2d(self, x, filters, k, name, stride=1, rate=1, padding='same', use_bias=False, out=None,
               pad_to=None, kernel_initializer='glorot_uniform'):
        """DeeplabConv2D (reference layers.py:14-21): glorot_uniform kernel (he_normal in the ResNet50 backbone), zero
        bias, l2(2e-5) on both"""
        H, W, cin = x.shape
        src = x
        if not isinstance(padding, str):
            src = self.add_layer(None, 'ZeroPadding2D', inbound=[x])
        Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
        layer = self.add_layer(name, 'Conv2D', (Ho, Wo, filters), inbound=[src])
        cdev = pad_to or filters
        # dense (k > 1) kernels are stored as the im2col GEMM operand [k*k*cin padded to a multiple of 4][cout]
        kp = (k * k * cin + self.align - 1) // self.align * self.align
        dev_shape = (k, k, cin, cdev) if k == 1 else (kp, cdev)
        if kernel_initializer == 'he_normal':
            init = lambda s: he_normal(self.rng, s, k * k * cin)
        else:
            init = lambda s: glorot_uniform(self.rng, s, k * k * cin, k * k * filters)
        wp = layer.add_param('kernel', (k, k, cin, filters), init, l2=L2_FACTOR, dev_shape=dev_shape)
        bp = None
        if use_bias:
            bp = layer.add_param('bias', (filters,), lambda s: np.zeros(s, np.float32), l2=L2_FACTOR,
                                 dev_shape=(cdev,))
        if out is None:
            out = self.new_tensor(Ho, Wo, cdev, name)
        assert (out.H, out.W, out.C) == (Ho, Wo, cdev), (name, (out.H, out.W, out.C), (Ho, Wo, cdev))
        # 1x1 stride 1 is a plain GEMM; everything else (3x3 stem, Xception's stride-2 1x1 shortcuts) goes through
        # im2col + the same GEMM
        kind = 'conv_pw' if (k == 1 and stride == 1) else 'conv_dense'
        col = self.new_tensor(Ho, Wo, kp, name + '_im2col') if kind == 'conv_dense' else None
        self.ops.append(Op(kind, name=name, layer=layer, x=x, w=wp, b=bp, out=out, k=k, stride=stride, rate=rate,
                           pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo, cin=cin, cout=cdev, bn=None, col=col, kp=kp))
        return Value(out, klayer=layer)

    def dwconv2d(self, x, k, name, stride=1, rate=1, padding='same', out=None):
        """DeeplabDepthwiseConv2D (reference layers.py:24-31); its kernel_regularizer never reaches the
        depthwise kernel in Keras -> no l2"""
        H, W, c = x.shape
        src = x
        if not isinstance(padding, str):
            src = self.add_layer(None, 'ZeroPadding2D', inbound=[x])
        Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
        layer = self.add_layer(name, 'DepthwiseConv2D', (Ho, Wo, c), inbound=[src])
        wp = layer.add_param('depthwise_kernel', (k, k, c, 1),
                             lambda s: glorot_uniform(self.rng, s, k * k * c, k * k * 1), l2=0.0)
        if out is None:
            out = self.new_tensor(Ho, Wo, c, name)
        self.ops.append(Op('conv_dw', name=name, layer=layer, x=x, w=wp, out=out, k=k, stride=stride, rate=rate,
                           pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo, c=c, bn=None))
        return Value(out, klayer=layer)

    def batchnorm(self, z, name, eps=1e-3, momentum=0.99, group=None, goff=0):
        """CustomBatchNormalization (reference layers.py:63-70)"""
        assert z.is_plain, 'BatchNormalization expects a raw conv output'
        t = z.tensor
        layer = self.add_layer(name, 'BatchNormalization', (t.H, t.W, t.C), inbound=[z])
        c = t.C
        layer.add_param('gamma', (c,), lambda s: np.ones(s, np.float32))
        layer.add_param('beta', (c,), lambda s: np.zeros(s, np.float32))
        layer.add_param('moving_mean', (c,), lambda s: np.zeros(s, np.float32), trainable=False)
        layer.add_param('moving_variance', (c,), lambda s: np.ones(s, np.float32), trainable=False)
        if group is None:
            group, goff = self.new_group(c), 0
        for i in range(goff, goff + c):
            group.identity[i] = False
        bn = BNSpec(layer, c, eps, momentum, group, goff)
        bn.z = t
        self.bns.append(bn)
        # attach to the producing conv so that it emits the statistics
        prod = self.producer_of(t)
        assert prod is not None and prod.bn is None, name
        prod.bn = bn
        self.ops.append(Op('bn', name=name, layer=layer, bn=bn, z=t, producer=prod))
        return Value(t, group, goff, ACT_NONE, bn, klayer=layer)

    def producer_of(self, t):
        for op in reversed(self.ops):
            if getattr(op, 'out', None) is t:
                return op
        return None

    def activation(self, v, act, name=None, kind='ReLU'):
        layer = self.add_layer(name, kind, v.shape, inbound=[v])
        if act == ACT_HSWISH:
            # hard_swish(x) = Multiply()([Activation(hard_sigmoid)(x), x]) (deeplabv3p_mobilenetv3.py:102-103): two Keras layers
            layer = self.add_layer(None, 'Multiply', v.shape, inbound=[layer, v])
        if act == ACT_NONE:
            return v.from_layer(layer)
        if v.act == act and act in (ACT_RELU, ACT_RELU6):
            return v.from_layer(layer)    # idempotent (ReLU in front of a SepConv_BN whose input is already ReLU-ed)
        assert v.act == ACT_NONE, 'stacked activations need a materialised tensor'
        if v.bn is not None:
            assert v.bn.act in (ACT_NONE, act), 'one activation per BatchNormalization (materialise otherwise)'
            v.bn.act = act
            return Value(v.tensor, v.group, v.goff, act, v.bn, klayer=layer)
        out = Value(v.tensor, v.group, v.goff, act, None, klayer=layer)
        if v.group is None:
            # bare activation of a materialised tensor (Xception: ReLU in front of a SepConv_BN applied to a
            # residual sum; SE block activations).  Its consumers write d/d(act(T)) into a view buffer that
            # backward folds into T's gradient as g * act'(T).
            key = (v.tensor.id, act)
            if key not in self.act_views:
                vt = self.new_tensor(v.tensor.H, v.tensor.W, v.tensor.C, 'actview%d_%s' % (act, v.tensor.name))
                vt.grad_only = True
                self.act_views[key] = (v.tensor, act, vt)
            out.view_grad = self.act_views[key][2]
        return out

    def relu(self, v, name=None):
        return self.activation(v, ACT_RELU, name)

    def relu6(self, v, name=None):
        return self.activation(v, ACT_RELU6, name)

    def materialize(self, v, residual=None, dropout=0.0, name=None, out=None):
        """y = dropout(v) [+ residual]: the Add / Dropout layers and any place a real tensor is needed"""
        H, W, C = v.shape
        if out is None:
            out = self.new_tensor(H, W, C, name or ('mat_' + v.tensor.name))
        dname = None
        if dropout:
            dname = 'dropout_%d' % self.dropout_count
            self.dropout_count += 1
        self.ops.append(Op('materialize', name=name or out.name, x=v, r=residual, rate=dropout, dropout_name=dname,
                           out=out))
        return Value(out, klayer=v.klayer)      # not a Keras layer by itself: add() / dropout() name theirs

    def add(self, a, b, name=None, keras_inputs=None):
        """Add([a, b]) -- `b` is the BN output (lazy), `a` the shortcut; keras_inputs: the list as the reference passes it
        when that is [b, a] (the order decides ties in Keras' layer ordering)"""
        layer = self.add_layer(name, 'Add', a.shape, inbound=keras_inputs or [a, b])
        return self.materialize(b, residual=a, name=name).from_layer(layer)

    def dropout(self, v, rate, name=None):
        layer = self.add_layer(name, 'Dropout', v.shape, inbound=[v])
        return self.materialize(v, dropout=rate, name=name).from_layer(layer)

    def se_multiply(self, x, s, name=None):
        """Multiply([x, s]) with s (1,1,C) broadcast over the pixels (reference deeplabv3p_mobilenetv3.py:145)"""
        H, W, C = x.shape
        layer = self.add_layer(name, 'Multiply', (H, W, C), inbound=[x, s])
        out = self.new_tensor(H, W, C, name or ('semul_' + x.tensor.name))
        self.ops.append(Op('se_mul', name=out.name, x=x, s=s, out=out))
        return Value(out, klayer=layer)

    def maxpool2d(self, v, k, stride, pad, name=None, pad_name=None):
        """ZeroPadding2D(pad) + MaxPooling2D((k,k), strides) (reference deeplabv3p_resnet50.py:266-267)"""
        H, W, C = v.shape
        pt, pb, pl, pr = pad
        src = v
        if any(pad):
            src = self.add_layer(pad_name, 'ZeroPadding2D', inbound=[v])
        Ho, Wo = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
        layer = self.add_layer(name, 'MaxPooling2D', (Ho, Wo, C), inbound=[src])
        out = self.new_tensor(Ho, Wo, C, layer.name)
        self.ops.append(Op('maxpool', name=layer.name, x=v, out=out, k=k, stride=stride, pad_t=pt, pad_l=pl, Ho=Ho, Wo=Wo))
        return Value(out, klayer=layer)

    def global_avgpool(self, v, name=None, kind='AveragePooling2D'):
        H, W, C = v.shape
        layer = self.add_layer(name, kind, (1, 1, C), inbound=[v])
        out = self.new_tensor(1, 1, C, 'pool_' + v.tensor.name)
        self.ops.append(Op('gap', name=out.name, x=v, out=out))
        return Value(out, klayer=layer)

    def passthrough(self, v, kind, output_shape=None, name=None):
        """a Keras layer that moves no data here (Reshape of a pooled vector, pred_resize / Softmax of the head)"""
        return v.from_layer(self.add_layer(name, kind, output_shape, inbound=[v]))

    def resize(self, v, H, W, name, out=None):
        """Lambda(img_resize, bilinear) (reference layers.py:48-60).  From a 1x1 map the bilinear
        resize is a broadcast, which commutes with the lazy BN+activation -> stays lazy."""
        h, w, C = v.shape
        layer = self.add_layer(name, 'Lambda', (H, W, C), inbound=[v])
        if out is None:
            out = self.new_tensor(H, W, C, name)
        if h == 1 and w == 1:
            self.ops.append(Op('broadcast', name=name, x=v, out=out))
            return Value(out, v.group, v.goff, v.act, v.bn, klayer=layer)
        assert v.is_plain, 'bilinear resize needs a materialised input'
        self.ops.append(Op('resize', name=name, x=v, out=out))
        return Value(out, klayer=layer)

    def concat_buffer(self, H, W, channels, name):
        """allocate the Concatenate target up front: returns (slices, coefficient group)"""
        total = sum(channels)
        base = self.new_tensor(H, W, total, name)
        group = self.new_group(total)
        slices, off = [], 0
        for c in channels:
            slices.append((base.slice(off, c), off))
            off += c
        return base, slices, group

    def concat_value(self, base, group, act, inputs, name=None):
        """`inputs`: the branch values in the order of the reference's Concatenate([...]) call"""
        layer = self.add_layer(name, 'Concatenate', (base.H, base.W, base.C), inbound=inputs)
        return Value(base, group, 0, act, None, klayer=layer)

    def keras_layer_order(self, output_layer=None):
        """`model.layers` as Keras builds it for a functional model (keras/engine/functional.py `_map_graph_network`):
        layers sorted by DECREASING depth -- the length of the longest path from the layer to the output -- and, inside
        one depth, by the order in which a depth-first walk from the output (inputs of a merge layer in list order) first
        meets them.  Creation order and this order differ wherever the graph branches: the ASPP branches, the Xception
        shortcut convs, the decoder's skip projection.  `save_weights` writes and `load_weights(by_name=False)` pairs
        layers in THIS order (the i-th layer with weights of the file feeds the i-th layer with weights of the model)."""
        out = output_layer or self.layers[-1]
        index, depth, post = {}, {}, []

        def walk(root):
            # iterative DFS: pre-order index at first visit, post-order list for the depth pass
            stack = [(root, 0)]
            index[root] = len(index)
            while stack:
                layer, i = stack.pop()
                if i < len(layer.inbound):
                    stack.append((layer, i + 1))
                    nxt = layer.inbound[i]
                    if nxt not in index:
                        index[nxt] = len(index)
                        stack.append((nxt, 0))
                else:
                    post.append(layer)
        walk(out)
        for layer in reversed(post):               # consumers before producers
            d = depth.setdefault(layer, 0)
            for src in layer.inbound:
                depth[src] = max(depth.get(src, 0), d + 1)
        by_depth = {}
        for layer, d in depth.items():
            by_depth.setdefault(d, []).append(layer)
        order = []
        for d in sorted(by_depth, reverse=True):
            order.extend(sorted(by_depth[d], key=lambda l: index[l]))
        return order

    # ---- finishing --------------------------------------------------------------------
    def init_weights(self):
        for layer in self.layers:
            for p in layer.params:
                if p.value is None:
                    p.value = np.asarray(p.init(p.shape), dtype=np.float32)

    def all_params(self):
        return [p for l in self.layers for p in l.params]

    def count_params(self, trainable=True):
        return sum(p.size for p in self.all_params() if p.weight_trainable == trainable)
