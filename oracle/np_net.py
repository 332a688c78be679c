"""CPU oracle: straight-line NumPy restatement of the reference's DeepLabV3+ graphs with a
minimal reverse-mode tape (forward, backward, SGD step).

TEST INFRASTRUCTURE ONLY (see oracle/np_ops.py header).  PARITY UNPINNED (TensorFlow not
importable).  Graph topology IS pinned: parameter counts are checked against the
reference's README.md:312-317 table in tests/test_oracle_topology.py, and the MobileNetV2 body + ASPP-Lite head agree layer by
layer with an unrelated third implementation (HuggingFace transformers' TF-slim ports, tests/test_oracle_vs_transformers.py).

Each builder cites the reference function it follows.  Parameter names follow the Keras layer
names of the reference so that weights can be exchanged by name with the HIP implementation:
  Conv2D            <layer>/kernel (kh,kw,Cin,Cout) [, <layer>/bias]
  DepthwiseConv2D   <layer>/depthwise_kernel (kh,kw,C,1)
  BatchNormalization<layer>/gamma, /beta, /moving_mean, /moving_variance
"""
import math
import numpy as np
from . import np_ops as O


class Var:
    __slots__ = ('v', 'g', 'tag', 'raw')

    def __init__(self, v):
        self.v = v
        self.g = None
        self.tag = None

    def acc(self, g):
        self.g = g if self.g is None else self.g + g


class Net:
    """Parameter store + tape.  One instance == one model replica."""

    def __init__(self, dtype=np.float64, seed=0):
        self.dtype = dtype
        self.rng = np.random.default_rng(seed)
        self.params = {}        # name -> ndarray (trainable + BN moving stats)
        self.order = []         # creation order (Keras topological weight order)
        self.trainable = {}     # name -> bool (False for moving stats)
        self.l2 = {}            # name -> l2 coefficient
        self.layer_trainable = {}  # layer name -> bool (freeze_level support)
        self.grads = {}
        self.tape = []
        self.training = True
        self.moving_updates = {}
        self.dropout_masks = {}  # layer name -> keep mask (injected); None -> no dropout
        self.taps = {}          # optional named intermediate activations
        self.reg_loss = 0.0
        # optional {BN layer name: act'(u) array}: ReLU-type activations are not differentiable at 0, so a
        # float32 implementation and this float64 oracle can disagree on the branch taken by elements
        # within rounding distance of the kink.  Parity tests inject the branch pattern of the
        # implementation under test so that the comparison is well-conditioned.
        self.act_derivs = {}
        # ... and, in creation order, the derivative arrays of bare activations applied to tensors that are
        # not BatchNormalization outputs (residual sums, SE-block convs)
        self.act_derivs_seq = []
        self._seq_pos = 0
        self.flip_count = 0
        self.flip_total = 0
        # mixed precision (train.py:37-46, bf16 on MI355X): with bf16 = True every layer OUTPUT that the HIP path stores or
        # forms in a consumer's prologue is rounded to bfloat16 -- conv / depthwise outputs, act(BN(z)) (the BatchNorm output
        # itself only where no activation follows), Add / Multiply / pooling / resize results, the conv kernels as the
        # matrix cores read them; statistics, biases, the logits (`keep_f32`) and the whole backward stay in this net's
        # dtype.  Forward values then follow the device's rounding points; gradients are compared with a tolerance.
        self.bf16 = False
        self.record = None      # optional dict: raw conv / depthwise outputs by layer name, as THIS net computes them
        # optional {layer name: array}: after recording its own result, a conv / depthwise layer continues with the given
        # output instead ("teacher forcing" with the device's stored tensors).  Two bf16 computations of one model drift
        # apart after the first element that rounds the other way (BatchNorm spreads it over the channel); forcing keeps
        # this oracle on the device's trajectory so that EVERY layer is compared on identical inputs and the backward
        # pass is linearised at the device's own activations.
        self.force = None
        # the same for the BACKWARD pass: {layer name: d loss / d (raw conv output) as the device stored it}.  A conv / depthwise
        # layer's backward first records the gradient THIS net computed for its output (record_grad), then continues with the
        # device's: its weight gradient and everything up to the next conv outputs (data gradient, BatchNorm backward with the
        # activation derivative, Add / concat / resize / pooling transposes) are then compared segment by segment on the
        # device's own inputs, instead of through a hundred layers of accumulated bf16 rounding.
        self.force_grad = None
        self.record_grad = None
        # optional dict: for every BatchNorm (dgamma, dbeta) the l2 norm of the terms that were summed (||g' xhat||, ||g'||).
        # A beta in front of a conv + BatchNorm pair has an exactly zero gradient (the next BatchNorm removes the shift), so
        # what a run computes for it is the rounding noise of its terms: this is the scale that noise is measured against.
        self.grad_term_norm = None
        # optional SyncBatchNorm: (all_reduce_sum(ndarray) -> ndarray, world_size).  Statistics are
        # summed over ranks in forward (sum x, sum x^2, count) and backward (sum dy, sum dy*xhat); the
        # parameter gradients stay local and are averaged with all other gradients (README.md:38,
        # layers.py:63-70; the protocol the HIP executor implements with RCCL).
        self.sync = None
        # which batch variance feeds the moving average: 'biased' (Keras SyncBatchNormalization, non-fused path) or 'unbiased'
        # (fused BatchNormalization, Bessel's correction) -- see np_ops.bn_moving_variance_of
        self.bn_moving_variance = 'biased'

    # ---- parameters -------------------------------------------------------------------
    def param(self, name, shape, init, trainable=True, l2=0.0):
        if name not in self.params:
            self.params[name] = init(shape).astype(self.dtype)
            self.order.append(name)
            self.trainable[name] = trainable
            self.l2[name] = l2
        return self.params[name]

    def layer_is_trainable(self, layer):
        return self.layer_trainable.get(layer, True)

    def q(self, a):
        return O.bf16_round(a) if self.bf16 else a

    def n_params(self, trainable=True):
        return sum(int(np.prod(self.params[n].shape)) for n in self.order if self.trainable[n] == trainable)

    # ---- tape -------------------------------------------------------------------------
    def begin(self, training=True):
        self.tape = []
        self.grads = {}
        self.moving_updates = {}
        self.training = training
        self.taps = {}
        self.reg_loss = 0.0
        self._seq_pos = 0
        self.flip_count = 0       # activation elements whose injected branch differs from the oracle's own
        self.flip_total = 0

    def backward(self):
        for f in reversed(self.tape):
            f()

    def acc_grad(self, name, g):
        self.grads[name] = g if name not in self.grads else self.grads[name] + g

    def _teacher_grad(self, name, y):
        """record_grad / force_grad at a conv output (see __init__)"""
        if self.record_grad is not None:
            self.record_grad[name] = y.g
        if self.force_grad is not None and name in self.force_grad:
            y.g = np.asarray(self.force_grad[name], dtype=self.dtype).reshape(y.g.shape)

    # ---- layers -----------------------------------------------------------------------
    def conv2d(self, x, filters, k, name, stride=1, rate=1, padding='same', use_bias=False, he_normal=False, keep_f32=False):
        """DeeplabConv2D (layers.py:14-21): glorot_uniform kernel (he_normal in ResNet50), zero bias, l2(2e-5) on both."""
        cin = x.v.shape[-1]
        init = ((lambda s: O.he_normal(self.rng, s, k * k * cin)) if he_normal else
                (lambda s: O.glorot_uniform(self.rng, s, k * k * cin, k * k * filters)))
        w = self.param(name + '/kernel', (k, k, cin, filters), init, l2=O.L2_FACTOR)
        b = self.param(name + '/bias', (filters,), np.zeros, l2=O.L2_FACTOR) if use_bias else None
        wq = self.q(w)
        yv = O.conv2d_fwd(x.v, wq, stride, rate, padding, b)
        y = Var(yv if keep_f32 else self.q(yv))
        if self.record is not None:
            self.record[name] = y.v
        if self.force is not None and name in self.force:
            y.v = np.asarray(self.force[name], dtype=self.dtype).reshape(y.v.shape)
        need_gx = True

        def bwd():
            if y.g is None:
                return
            self._teacher_grad(name, y)
            gx, gw, gb = O.conv2d_bwd(x.v, wq, y.g, stride, rate, padding, need_gx)
            self.acc_grad(name + '/kernel', gw)
            if use_bias:
                self.acc_grad(name + '/bias', gb)
                if self.grad_term_norm is not None:      # (a bias in front of a BatchNorm has an exactly zero gradient: noise scale)
                    self.grad_term_norm[name + '/bias'] = np.sqrt((np.asarray(y.g, np.float64).reshape(-1, filters) ** 2).sum(0))
            x.acc(gx)
        self.tape.append(bwd)
        return y

    def dwconv2d(self, x, k, name, stride=1, rate=1, padding='same'):
        """DeeplabDepthwiseConv2D (layers.py:24-31).  kernel_regularizer is NOT attached to the
        depthwise kernel by Keras -> l2 = 0."""
        c = x.v.shape[-1]
        w4 = self.param(name + '/depthwise_kernel', (k, k, c, 1),
                        lambda s: O.glorot_uniform(self.rng, s, k * k * c, k * k * 1), l2=0.0)
        w = self.q(w4[..., 0])
        y = Var(self.q(O.dwconv2d_fwd(x.v, w, stride, rate, padding)))
        if self.record is not None:
            self.record[name] = y.v
        if self.force is not None and name in self.force:
            y.v = np.asarray(self.force[name], dtype=self.dtype).reshape(y.v.shape)

        def bwd():
            if y.g is None:
                return
            self._teacher_grad(name, y)
            gx, gw = O.dwconv2d_bwd(x.v, w, y.g, stride, rate, padding)
            self.acc_grad(name + '/depthwise_kernel', gw[..., None])
            x.acc(gx)
        self.tape.append(bwd)
        return y

    def bn(self, x, name, eps=1e-3, momentum=0.99):
        """CustomBatchNormalization (layers.py:63-70).  A frozen (non-trainable) BN layer runs in
        inference mode (TF2 semantics)."""
        c = x.v.shape[-1]
        gamma = self.param(name + '/gamma', (c,), np.ones)
        beta = self.param(name + '/beta', (c,), np.zeros)
        mm = self.param(name + '/moving_mean', (c,), np.zeros, trainable=False)
        mv = self.param(name + '/moving_variance', (c,), np.ones, trainable=False)
        if self.training and self.layer_is_trainable(name) and self.sync is not None:
            return self._sync_bn(x, name, gamma, beta, mm, mv, eps, momentum)
        if self.training and self.layer_is_trainable(name):
            yv, cache, (bm, bv) = O.bn_train_fwd(x.v, gamma, beta, eps)
            self.moving_updates[name + '/moving_mean'] = O.bn_moving_update(mm, bm, momentum)
            bv = O.bn_moving_variance_of(bv, x.v.size // c, self.bn_moving_variance)
            self.moving_updates[name + '/moving_variance'] = O.bn_moving_update(mv, bv, momentum)
            y = Var(self.q(yv))
            if self.bf16:
                y.raw = yv          # an activation that follows rounds act(BN(z)) once, like the kernels' prologue
            y.tag = name

            def bwd():
                if y.g is None:
                    return
                gx, gg, gb = O.bn_train_bwd(y.g, cache)
                self.acc_grad(name + '/gamma', gg)
                self.acc_grad(name + '/beta', gb)
                if self.grad_term_norm is not None:
                    g2 = np.asarray(y.g, np.float64).reshape(-1, c)
                    self.grad_term_norm[name + '/beta'] = np.sqrt((g2 ** 2).sum(0))
                    self.grad_term_norm[name + '/gamma'] = np.sqrt(((g2 * np.asarray(cache[0], np.float64).reshape(-1, c)) ** 2).sum(0))
                x.acc(gx)
        else:
            yv = O.bn_infer_fwd(x.v, gamma, beta, mm, mv, eps)
            y = Var(self.q(yv))
            if self.bf16:
                y.raw = yv
            y.tag = name
            scale = gamma / np.sqrt(mv + eps)

            def bwd():
                if y.g is None:
                    return
                xh = (x.v - mm) / np.sqrt(mv + eps)
                self.acc_grad(name + '/gamma', (y.g * xh).reshape(-1, c).sum(0))
                self.acc_grad(name + '/beta', y.g.reshape(-1, c).sum(0))
                x.acc(y.g * scale)
        self.tape.append(bwd)
        return y

    def _sync_bn(self, x, name, gamma, beta, mm, mv, eps, momentum):
        allreduce, world = self.sync
        c = x.v.shape[-1]
        x2 = x.v.reshape(-1, c)
        m_local = x2.shape[0]
        sums = allreduce(np.concatenate([x2.sum(0), (x2 ** 2).sum(0)]))
        m = m_local * world
        mean = sums[:c] / m
        var = np.maximum(sums[c:] / m - mean ** 2, 0.0)
        invstd = 1.0 / np.sqrt(var + eps)
        xhat = (x.v - mean) * invstd
        y = Var(xhat * gamma + beta)
        y.tag = name
        self.moving_updates[name + '/moving_mean'] = O.bn_moving_update(mm, mean, momentum)
        self.moving_updates[name + '/moving_variance'] = O.bn_moving_update(
            mv, O.bn_moving_variance_of(var, m, self.bn_moving_variance), momentum)

        def bwd():
            if y.g is None:
                return
            g2 = y.g.reshape(-1, c)
            sdy, sdyx = g2.sum(0), (g2 * xhat.reshape(-1, c)).sum(0)
            self.acc_grad(name + '/gamma', sdyx)
            self.acc_grad(name + '/beta', sdy)
            tot = allreduce(np.concatenate([sdy, sdyx]))
            x.acc((gamma * invstd) * (y.g - tot[:c] / m - xhat * (tot[c:] / m)))
        self.tape.append(bwd)
        return y

    def act(self, x, kind):
        # bf16: the device forms act(z * scale + shift) from the UNROUNDED affine value (fp32 in the consumer's prologue) and
        # evaluates the activation's derivative there too -- the BatchNorm output itself is never stored -- so both directions
        # use x.raw here; through the rounded x.v every pre-activation within 2^-9 of a kink (0, 6, +-3) would take the other
        # branch of the derivative (an O(1) error in that element's gradient, ~1e-3 of all elements)
        xin = getattr(x, 'raw', x.v) if self.bf16 else x.v
        y = Var(self.q(O.act_fwd(xin, kind)))
        y.tag = ('act',)
        deriv = None
        if isinstance(x.tag, str):
            deriv = self.act_derivs.get(x.tag)
        elif x.tag is None and self.act_derivs_seq:
            deriv = self.act_derivs_seq[self._seq_pos]
            self._seq_pos += 1

        if deriv is not None:
            # how much the injection changes: elements whose branch differs from this oracle's own derivative
            own = O.act_bwd(xin, np.ones_like(x.v), kind)
            self.flip_count += int(np.count_nonzero(np.abs(own - deriv) > 1e-3))      # a flipped branch moves the derivative by O(1)
            self.flip_total += int(own.size)

        def bwd():
            if y.g is not None:
                if deriv is not None:
                    x.acc(y.g * deriv)
                else:
                    x.acc(O.act_bwd(xin, y.g, kind))
        self.tape.append(bwd)
        return y

    def relu(self, x):
        return self.act(x, O.ACT_RELU)

    def relu6(self, x):
        return self.act(x, O.ACT_RELU6)

    def add(self, a, b):
        y = Var(self.q(a.v + b.v))

        def bwd():
            if y.g is not None:
                a.acc(y.g)
                b.acc(y.g)
        self.tape.append(bwd)
        return y

    def mul_bcast(self, x, s):
        """x (N,H,W,C) * s (N,1,1,C)   (SE block Multiply, deeplabv3p_mobilenetv3.py:145)"""
        y = Var(self.q(x.v * s.v))

        def bwd():
            if y.g is not None:
                x.acc(y.g * s.v)
                s.acc((y.g * x.v).sum(axis=(1, 2), keepdims=True))
        self.tape.append(bwd)
        return y

    def concat(self, xs):
        y = Var(np.concatenate([x.v for x in xs], axis=-1))
        sizes = [x.v.shape[-1] for x in xs]

        def bwd():
            if y.g is None:
                return
            o = 0
            for x, s in zip(xs, sizes):
                x.acc(y.g[..., o:o + s])
                o += s
        self.tape.append(bwd)
        return y

    def maxpool2d(self, x, k, stride, pad):
        """ZeroPadding2D(pad) + MaxPooling2D((k,k), strides) (deeplabv3p_resnet50.py:266-267)"""
        yv, arg = O.maxpool2d_fwd(x.v, k, stride, pad)
        y = Var(yv)
        shape = x.v.shape

        def bwd():
            if y.g is not None:
                x.acc(O.maxpool2d_bwd(y.g, arg, shape, k, stride, pad))
        self.tape.append(bwd)
        return y

    def global_avgpool(self, x):
        H, W = x.v.shape[1:3]
        y = Var(self.q(O.global_avgpool_fwd(x.v)))

        def bwd():
            if y.g is not None:
                x.acc(O.global_avgpool_bwd(y.g, H, W))
        self.tape.append(bwd)
        return y

    def resize(self, x, out_h, out_w, keep_f32=False):
        """img_resize (layers.py:48-60) bilinear"""
        H, W = x.v.shape[1:3]
        yv = O.resize_bilinear_fwd(x.v, out_h, out_w)
        y = Var(yv if keep_f32 else self.q(yv))

        def bwd():
            if y.g is not None:
                x.acc(O.resize_bilinear_bwd(y.g, H, W))
        self.tape.append(bwd)
        return y

    def dropout(self, x, name, rate=0.5):
        mask = self.dropout_masks.get(name)
        if not self.training or mask is None:
            return x
        mask = mask.astype(self.dtype)
        y = Var(O.dropout_fwd(x.v, mask, rate))

        def bwd():
            if y.g is not None:
                x.acc(O.dropout_bwd(y.g, mask, rate))
        self.tape.append(bwd)
        return y

    def tap(self, name, x):
        self.taps[name] = x
        return x

    # ---- blocks (deeplabv3p/models/layers.py) -------------------------------------------
    def sepconv_bn(self, x, filters, prefix, stride=1, k=3, rate=1, depth_activation=False, eps=1e-3):
        """SepConv_BN (layers.py:74-111)"""
        if stride == 1:
            padding = 'same'
        else:
            k_eff = k + (k - 1) * (rate - 1)
            pad_total = k_eff - 1
            pb = pad_total // 2
            pe = pad_total - pb
            padding = (pb, pe, pb, pe)        # ZeroPadding2D((pad_beg,pad_end)) on H and W, then VALID
        if not depth_activation:
            x = self.relu(x)
        x = self.dwconv2d(x, k, prefix + '_depthwise', stride, rate, padding)
        x = self.bn(x, prefix + '_depthwise_BN', eps)
        if depth_activation:
            x = self.relu(x)
        x = self.conv2d(x, filters, 1, prefix + '_pointwise')
        x = self.bn(x, prefix + '_pointwise_BN', eps)
        if depth_activation:
            x = self.relu(x)
        return x

    def aspp_image_branch(self, x):
        """layers.py:131-138: AveragePooling2D(full map) -> 1x1(256) -> BN(1e-5) -> ReLU -> bilinear up"""
        H, W = x.v.shape[1:3]
        b4 = self.global_avgpool(x)
        b4 = self.conv2d(b4, 256, 1, 'image_pooling')
        b4 = self.bn(b4, 'image_pooling_BN', 1e-5)
        b4 = self.relu(b4)
        return self.resize(b4, H, W)

    def aspp_block(self, x, OS):
        """ASPP_block (layers.py:114-163)"""
        rates = {8: (12, 24, 36), 16: (6, 12, 18), 32: (3, 6, 9)}[OS]
        b4 = self.aspp_image_branch(x)
        b0 = self.relu(self.bn(self.conv2d(x, 256, 1, 'aspp0'), 'aspp0_BN', 1e-5))
        b1 = self.sepconv_bn(x, 256, 'aspp1', rate=rates[0], depth_activation=True, eps=1e-5)
        b2 = self.sepconv_bn(x, 256, 'aspp2', rate=rates[1], depth_activation=True, eps=1e-5)
        b3 = self.sepconv_bn(x, 256, 'aspp3', rate=rates[2], depth_activation=True, eps=1e-5)
        x = self.concat([b4, b0, b1, b2, b3])
        x = self.relu(self.bn(self.conv2d(x, 256, 1, 'concat_projection'), 'concat_projection_BN', 1e-5))
        return self.dropout(x, 'aspp_dropout', 0.5)

    def aspp_lite_block(self, x):
        """ASPP_Lite_block (layers.py:166-196)"""
        b4 = self.aspp_image_branch(x)
        b0 = self.relu(self.bn(self.conv2d(x, 256, 1, 'aspp0'), 'aspp0_BN', 1e-5))
        x = self.concat([b4, b0])
        x = self.relu(self.bn(self.conv2d(x, 256, 1, 'concat_projection'), 'concat_projection_BN', 1e-5))
        return self.dropout(x, 'aspp_dropout', 0.5)

    def decoder_block(self, x, skip):
        """Decoder_block (layers.py:199-219)"""
        H, W = skip.v.shape[1:3]
        x = self.resize(x, H, W)
        s = self.relu(self.bn(self.conv2d(skip, 48, 1, 'feature_projection0'), 'feature_projection0_BN', 1e-5))
        x = self.concat([x, s])
        x = self.sepconv_bn(x, 256, 'decoder_conv0', depth_activation=True, eps=1e-5)
        x = self.sepconv_bn(x, 256, 'decoder_conv1', depth_activation=True, eps=1e-5)
        return x

    # ---- MobileNetV2 (deeplabv3p/models/deeplabv3p_mobilenetv2.py) ---------------------
    def mnv2_block(self, x, expansion, stride, filters, block_id, skip_connection, rate=1):
        """_inverted_res_block (deeplabv3p_mobilenetv2.py:38-74), alpha=1"""
        cin = x.v.shape[-1]
        pw = make_divisible(int(filters * 1.0), 8)
        inputs = x
        if block_id:
            prefix = 'expanded_conv_{}_'.format(block_id)
            x = self.conv2d(x, expansion * cin, 1, prefix + 'expand')
            x = self.bn(x, prefix + 'expand_BN', 1e-3, 0.999)
            x = self.relu6(x)
        else:
            prefix = 'expanded_conv_'
        x = self.dwconv2d(x, 3, prefix + 'depthwise', stride, rate, 'same')
        x = self.bn(x, prefix + 'depthwise_BN', 1e-3, 0.999)
        x = self.relu6(x)
        x = self.conv2d(x, pw, 1, prefix + 'project')
        x = self.bn(x, prefix + 'project_BN', 1e-3, 0.999)
        if skip_connection:
            x = self.add(inputs, x)
        return x

    def mobilenetv2_body(self, x, OS):
        """MobileNetV2_body (deeplabv3p_mobilenetv2.py:77-199); the Conv_1 tail (:166-175) is a dead
        branch and is not built."""
        s16, r16, s32, r32 = os_table(OS)
        x = self.conv2d(x, 32, 3, 'Conv', stride=2, padding='same')
        x = self.relu6(self.bn(x, 'Conv_BN', 1e-3, 0.999))
        x = self.mnv2_block(x, 1, 1, 16, 0, False)
        x = self.mnv2_block(x, 6, 2, 24, 1, False)
        x = self.mnv2_block(x, 6, 1, 24, 2, True)
        skip = x
        x = self.mnv2_block(x, 6, 2, 32, 3, False)
        x = self.mnv2_block(x, 6, 1, 32, 4, True)
        x = self.mnv2_block(x, 6, 1, 32, 5, True)
        x = self.mnv2_block(x, 6, s16, 64, 6, False)
        x = self.mnv2_block(x, 6, 1, 64, 7, True, rate=r16)
        x = self.mnv2_block(x, 6, 1, 64, 8, True, rate=r16)
        x = self.mnv2_block(x, 6, 1, 64, 9, True, rate=r16)
        x = self.mnv2_block(x, 6, 1, 96, 10, False, rate=r16)
        x = self.mnv2_block(x, 6, 1, 96, 11, True, rate=r16)
        x = self.mnv2_block(x, 6, 1, 96, 12, True, rate=r16)
        x = self.mnv2_block(x, 6, s32, 160, 13, False, rate=r16)
        x = self.mnv2_block(x, 6, 1, 160, 14, True, rate=r32)
        x = self.mnv2_block(x, 6, 1, 160, 15, True, rate=r32)
        x = self.mnv2_block(x, 6, 1, 320, 16, False, rate=r32)
        return x, skip

    # ---- Xception (deeplabv3p/models/deeplabv3p_xception.py) ---------------------------
    def xception_block(self, x, depth_list, prefix, skip_type, stride, rate=1, depth_activation=False,
                       return_skip=False):
        """_xception_block (deeplabv3p_xception.py:57-93)"""
        inputs = x
        res = x
        skip = None
        for i in range(3):
            res = self.sepconv_bn(res, depth_list[i], prefix + '_separable_conv{}'.format(i + 1),
                                  stride=stride if i == 2 else 1, rate=rate,
                                  depth_activation=depth_activation)
            if i == 1:
                skip = res
        if skip_type == 'conv':
            # _conv2d_same with kernel_size=1 (deeplabv3p_xception.py:25-54): k_eff-1 = 0 -> no pad,
            # stride-s 1x1 VALID conv == subsample then GEMM
            sc = self.conv2d(inputs, depth_list[-1], 1, prefix + '_shortcut', stride=stride,
                             padding='same' if stride == 1 else (0, 0, 0, 0))
            sc = self.bn(sc, prefix + '_shortcut_BN', 1e-3, 0.99)
            out = self.add(res, sc)
        elif skip_type == 'sum':
            out = self.add(res, inputs)
        else:
            out = res
        return (out, skip) if return_skip else out

    def xception_body(self, x, OS):
        """Xception_body (deeplabv3p_xception.py:96-163)"""
        s16, r16, s32, r32 = os_table(OS)
        x = self.conv2d(x, 32, 3, 'entry_flow_conv1_1', stride=2, padding='same')
        x = self.relu(self.bn(x, 'entry_flow_conv1_1_BN'))
        x = self.conv2d(x, 64, 3, 'entry_flow_conv1_2', stride=1, padding='same')
        x = self.relu(self.bn(x, 'entry_flow_conv1_2_BN'))
        x = self.xception_block(x, [128, 128, 128], 'entry_flow_block1', 'conv', 2)
        x, skip = self.xception_block(x, [256, 256, 256], 'entry_flow_block2', 'conv', 2, return_skip=True)
        x = self.xception_block(x, [728, 728, 728], 'entry_flow_block3', 'conv', s16)
        for i in range(16):
            x = self.xception_block(x, [728, 728, 728], 'middle_flow_unit_{}'.format(i + 1), 'sum', 1, rate=r16)
        x = self.xception_block(x, [728, 1024, 1024], 'exit_flow_block1', 'conv', s32, rate=r16)
        x = self.xception_block(x, [1536, 1536, 2048], 'exit_flow_block2', 'none', 1, rate=r32,
                                depth_activation=True)
        return x, skip

    # ---- MobileNetV3-Large (deeplabv3p/models/deeplabv3p_mobilenetv3.py) ------------------
    def se_block(self, x, filters, se_ratio, prefix):
        """_se_block (deeplabv3p_mobilenetv3.py:122-146)"""
        s = self.global_avgpool(x)
        s = self.conv2d(s, make_divisible(filters * se_ratio, 8), 1, prefix + 'squeeze_excite/Conv', use_bias=True)
        s = self.relu(s)
        s = self.conv2d(s, filters, 1, prefix + 'squeeze_excite/Conv_1', use_bias=True)
        s = self.act(s, O.ACT_HSIGMOID)
        return self.mul_bcast(x, s)

    def mnv3_block(self, x, expansion, filters, k, stride, se_ratio, activation, block_id,
                   skip_connection=False, rate=1):
        """_inverted_res_block (deeplabv3p_mobilenetv3.py:149-201)"""
        shortcut = x
        prefix = 'expanded_conv/'
        cin = x.v.shape[-1]
        if block_id:
            prefix = 'expanded_conv_{}/'.format(block_id)
            x = self.conv2d(x, make_divisible(cin * expansion, 8), 1, prefix + 'expand')
            x = self.bn(x, prefix + 'expand/BatchNorm', 1e-3, 0.999)
            x = self.act(x, activation)
        x = self.dwconv2d(x, k, prefix + 'depthwise/Conv', stride, rate, 'same')
        x = self.bn(x, prefix + 'depthwise/BatchNorm', 1e-3, 0.999)
        x = self.act(x, activation)
        if se_ratio:
            x = self.se_block(x, make_divisible(cin * expansion, 8), se_ratio, prefix)
        x = self.conv2d(x, filters, 1, prefix + 'project')
        x = self.bn(x, prefix + 'project/BatchNorm', 1e-3, 0.999)
        if skip_connection:
            x = self.add(shortcut, x)
        return x

    def mobilenetv3large_body(self, x, OS):
        """MobileNetV3 stem (deeplabv3p_mobilenetv3.py:343-355) + MobileNetV3Large.stack_fn (:551-593),
        alpha=1.0, kernel=5, activation=hard_swish, se_ratio=0.25"""
        s16, r16, s32, r32 = os_table(OS)
        H, W = x.v.shape[1:3]
        # ZeroPadding2D(correct_pad(x,3)) + 3x3 s2 VALID  (:343-350)
        adj_h, adj_w = 1 - H % 2, 1 - W % 2
        pads = (1 - adj_h, 1, 1 - adj_w, 1)
        x = self.conv2d(x, 16, 3, 'Conv', stride=2, padding=pads)
        x = self.bn(x, 'Conv/BatchNorm', 1e-3, 0.999)
        x = self.act(x, O.ACT_HSWISH)
        RE, HS = O.ACT_RELU, O.ACT_HSWISH
        d = lambda v: make_divisible(v, 8)
        se, k = 0.25, 5
        x = self.mnv3_block(x, 1, d(16), 3, 1, None, RE, 0, True)
        x = self.mnv3_block(x, 4, d(24), 3, 2, None, RE, 1, False)
        x = self.mnv3_block(x, 3, d(24), 3, 1, None, RE, 2, True)
        skip = x
        x = self.mnv3_block(x, 3, d(40), k, 2, se, RE, 3, False)
        x = self.mnv3_block(x, 3, d(40), k, 1, se, RE, 4, True)
        x = self.mnv3_block(x, 3, d(40), k, 1, se, RE, 5, True)
        x = self.mnv3_block(x, 6, d(80), 3, s16, None, HS, 6, False)
        x = self.mnv3_block(x, 2.5, d(80), 3, 1, None, HS, 7, True, rate=r16)
        x = self.mnv3_block(x, 2.3, d(80), 3, 1, None, HS, 8, True, rate=r16)
        x = self.mnv3_block(x, 2.3, d(80), 3, 1, None, HS, 9, True, rate=r16)
        x = self.mnv3_block(x, 6, d(112), 3, 1, se, HS, 10, False, rate=r16)
        x = self.mnv3_block(x, 6, d(112), 3, 1, se, HS, 11, True, rate=r16)
        x = self.mnv3_block(x, 6, d(160), k, s32, se, HS, 12, False, rate=r16)
        x = self.mnv3_block(x, 6, d(160), k, 1, se, HS, 13, True, rate=r32)
        x = self.mnv3_block(x, 6, d(160), k, 1, se, HS, 14, True, rate=r32)
        return x, skip


def make_divisible(v, divisor=8, min_value=None):
    """_make_divisible (deeplabv3p_mobilenetv2.py:28-35) == _depth (deeplabv3p_mobilenetv3.py:112-119)"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def os_table(OS):
    """(origin_os16_stride, origin_os16_block_rate, origin_os32_stride, origin_os32_block_rate)
    deeplabv3p_mobilenetv2.py:82-98 == deeplabv3p_xception.py:101-117"""
    if OS == 8:
        return 1, 2, 1, 4
    if OS == 16:
        return 2, 1, 1, 2
    if OS == 32:
        return 2, 1, 2, 1
    raise ValueError('invalid output stride', OS)


MODEL_TYPES = ('mobilenetv2', 'mobilenetv2_lite', 'xception', 'mobilenetv3large', 'mobilenetv3large_lite',
               'mobilenetv3small', 'mobilenetv3small_lite', 'resnet50')


def _resnet50_body(self, x, OS):
    """deeplabv3p_resnet50.py: identity_block :32-75, conv_block :78-139, body :262-292 (stem conv1_pad + 7x7 s2 valid,
    pool1_pad + MaxPooling2D(3, 2), stages 2-5 with the output-stride table :208-226); every conv has a bias and is
    followed by BatchNormalization with the Keras defaults"""
    s16, r16, s32, r32 = os_table(OS)

    def cbn(x, f, k, cn, bn, stride=1, rate=1, padding='same'):
        x = self.conv2d(x, f, k, cn, stride=stride, rate=rate, padding=padding, use_bias=True, he_normal=True)
        return self.bn(x, bn, 1e-3, 0.99)

    def identity(x, filters, stage, block, rate=1):
        cn, bn = 'res%d%s_branch' % (stage, block), 'bn%d%s_branch' % (stage, block)
        y = self.relu(cbn(x, filters[0], 1, cn + '2a', bn + '2a'))
        y = self.relu(cbn(y, filters[1], 3, cn + '2b', bn + '2b', rate=rate))
        y = cbn(y, filters[2], 1, cn + '2c', bn + '2c')
        return self.relu(self.add(y, x))

    def conv(x, filters, stage, block, strides=2, rate=1):
        cn, bn = 'res%d%s_branch' % (stage, block), 'bn%d%s_branch' % (stage, block)
        y = self.relu(cbn(x, filters[0], 1, cn + '2a', bn + '2a', stride=strides, padding='valid'))
        y = self.relu(cbn(y, filters[1], 3, cn + '2b', bn + '2b', rate=rate))
        y = cbn(y, filters[2], 1, cn + '2c', bn + '2c')
        sc = cbn(x, filters[2], 1, cn + '1', bn + '1', stride=strides, padding='valid')
        return self.relu(self.add(y, sc))

    x = self.relu(cbn(x, 64, 7, 'conv1', 'bn_conv1', stride=2, padding=(3, 3, 3, 3)))
    x = self.maxpool2d(x, 3, 2, (1, 1, 1, 1))
    x = conv(x, [64, 64, 256], 2, 'a', strides=1)
    x = identity(x, [64, 64, 256], 2, 'b')
    x = identity(x, [64, 64, 256], 2, 'c')
    skip = x
    x = conv(x, [128, 128, 512], 3, 'a')
    for b in 'bcd':
        x = identity(x, [128, 128, 512], 3, b)
    x = conv(x, [256, 256, 1024], 4, 'a', strides=s16)
    for b in 'bcdef':
        x = identity(x, [256, 256, 1024], 4, b, rate=r16)
    x = conv(x, [512, 512, 2048], 5, 'a', strides=s32, rate=r16)
    for b in 'bc':
        x = identity(x, [512, 512, 2048], 5, b, rate=r32)
    return x, skip


Net.resnet50_body = _resnet50_body


def _mobilenetv3small_body(self, x, OS):
    """MobileNetV3 stem (deeplabv3p_mobilenetv3.py:343-355) + MobileNetV3Small.stack_fn (:469-499): alpha=1.0, kernel=5,
    activation=hard_swish, se_ratio=0.25; skip feature = block 0's output"""
    s16, r16, s32, r32 = os_table(OS)
    H, W = x.v.shape[1:3]
    adj_h, adj_w = 1 - H % 2, 1 - W % 2
    x = self.conv2d(x, 16, 3, 'Conv', stride=2, padding=(1 - adj_h, 1, 1 - adj_w, 1))
    x = self.bn(x, 'Conv/BatchNorm', 1e-3, 0.999)
    x = self.act(x, O.ACT_HSWISH)
    RE, HS = O.ACT_RELU, O.ACT_HSWISH
    d = lambda v: make_divisible(v, 8)
    se, k = 0.25, 5
    x = self.mnv3_block(x, 1, d(16), 3, 2, se, RE, 0, False)
    skip = x
    x = self.mnv3_block(x, 72. / 16, d(24), 3, 2, None, RE, 1, False)
    x = self.mnv3_block(x, 88. / 24, d(24), 3, 1, None, RE, 2, True)
    x = self.mnv3_block(x, 4, d(40), k, s16, se, HS, 3, False)
    x = self.mnv3_block(x, 6, d(40), k, 1, se, HS, 4, True, rate=r16)
    x = self.mnv3_block(x, 6, d(40), k, 1, se, HS, 5, True, rate=r16)
    x = self.mnv3_block(x, 3, d(48), k, 1, se, HS, 6, False, rate=r16)
    x = self.mnv3_block(x, 3, d(48), k, 1, se, HS, 7, True, rate=r16)
    x = self.mnv3_block(x, 6, d(96), k, s32, se, HS, 8, False, rate=r16)
    x = self.mnv3_block(x, 6, d(96), k, 1, se, HS, 9, True, rate=r32)
    x = self.mnv3_block(x, 6, d(96), k, 1, se, HS, 10, True, rate=r32)
    return x, skip


Net.mobilenetv3small_body = _mobilenetv3small_body


class OracleModel:
    """get_deeplabv3p_model (deeplabv3p/model.py:51-117) restated: backbone + ASPP(+decoder), the
    21-class stub head dropped at layers[-5] and replaced by conv_upsample (1x1 + bias) ->
    pred_resize (bilinear to the input size) -> [Reshape] -> Softmax('pred_mask')."""

    net_class = Net          # oracle/torch_net.py substitutes a torch-autograd implementation of the same layer set

    def __init__(self, model_type, num_classes, input_shape, output_stride, dtype=np.float64, seed=0,
                 freeze_level=0, bn_moving_variance='biased'):
        if model_type not in MODEL_TYPES:
            raise ValueError('This model type is not supported now')
        self.model_type = model_type
        self.num_classes = num_classes
        self.H, self.W = input_shape
        self.OS = output_stride
        self.net = self.net_class(dtype, seed)
        self.net.bn_moving_variance = bn_moving_variance
        self.velocity = {}
        self.freeze_level = freeze_level
        # materialise parameters with one dry forward at batch 1 on a small probe (shapes of the
        # parameters do not depend on the spatial size)
        probe = np.zeros((1, 33, 33, 3), dtype=np.float64)
        self._forward_graph(probe, 33, 33, training=False)
        self.backbone_param_names = list(self._backbone_names)
        if freeze_level in (1, 2):
            # model.py:106-110: freeze the backbone (1) or everything but the new head (2)
            for n in self.net.order:
                layer = n.rsplit('/', 1)[0]
                in_backbone = layer in self._backbone_layers
                if freeze_level == 2:
                    frozen = layer != 'conv_upsample'
                else:
                    frozen = in_backbone
                self.net.layer_trainable[layer] = not frozen

    def _forward_graph(self, x, H, W, training):
        net = self.net
        net.begin(training)
        xin = Var(net.q(x))
        n_before = len(net.order)
        if self.model_type in ('mobilenetv2', 'mobilenetv2_lite'):
            f, skip = net.mobilenetv2_body(xin, self.OS)
        elif self.model_type == 'xception':
            f, skip = net.xception_body(xin, self.OS)
        elif self.model_type == 'resnet50':
            f, skip = net.resnet50_body(xin, self.OS)
        elif self.model_type.startswith('mobilenetv3small'):
            f, skip = net.mobilenetv3small_body(xin, self.OS)
        else:
            f, skip = net.mobilenetv3large_body(xin, self.OS)
        if not hasattr(self, '_backbone_names'):
            self._backbone_names = net.order[n_before:]
            self._backbone_layers = {n.rsplit('/', 1)[0] for n in self._backbone_names}
        net.tap('backbone_out', f)
        if self.model_type.endswith('_lite'):      # Deeplabv3pLite*: ASPP-Lite, no decoder
            y = net.aspp_lite_block(f)
        else:
            y = net.aspp_block(f, self.OS)
            net.tap('aspp_out', y)
            y = net.decoder_block(y, skip)
        net.tap('head_in', y)
        y = net.conv2d(y, self.num_classes, 1, 'conv_upsample', use_bias=True, keep_f32=True)   # fp32 logits on the bf16 path
        net.tap('conv_upsample', y)
        logits = net.resize(y, H, W, keep_f32=True)            # 'pred_resize' (model.py:76)
        net.tap('pred_resize', logits)
        return logits

    # ---- public -----------------------------------------------------------------------
    def trainable_param_names(self):
        net = self.net
        return [n for n in net.order if net.trainable[n] and net.layer_is_trainable(n.rsplit('/', 1)[0])]

    def predict(self, x):
        """inference mode: BN moving stats, no dropout.  returns (logits, probs) as (N,H,W,C)"""
        logits = self._forward_graph(x.astype(self.net.dtype), x.shape[1], x.shape[2], training=False)
        return logits.v, O.softmax_fwd(logits.v)

    def forward_train(self, x, dropout_masks=None):
        self.net.dropout_masks = dropout_masks or {}
        logits = self._forward_graph(x.astype(self.net.dtype), x.shape[1], x.shape[2], training=True)
        return logits

    def loss_and_grads(self, x, labels, dropout_masks=None, ignore_index=255, loss=None, sample_weight=None):
        """labels: (N, H*W, 1) float class ids (data.py:39-41).  returns (total_loss, ce_loss, logits);
        gradients (data term only; the l2 term is applied inside sgd_step like Keras adds it to the
        loss) are left in self.net.grads."""
        net = self.net
        logits = self.forward_train(x, dropout_masks)
        N, H, W, C = logits.v.shape
        lab = labels.reshape(N, H, W)
        sw = None if sample_weight is None else np.asarray(sample_weight).reshape(N, H, W)
        ce, probs, dlogits = O.loss_fwd_bwd(logits.v, lab, loss, ignore_index, sw)  # loss: see np_ops.loss_fwd_bwd
        logits.g = dlogits
        net.backward()
        reg = 0.0
        for n in self.trainable_param_names():
            if net.l2[n]:
                reg += net.l2[n] * float((net.params[n].astype(np.float64) ** 2).sum())
        self.last_probs = probs
        return ce + reg, ce, logits.v

    def sgd_step(self, lr=1e-2, momentum=0.9, optimizer='sgd'):
        """one optimiser step: 'sgd' (momentum), 'adam', 'rmsprop' (common/model_utils.py:118-124)"""
        net = self.net
        self.iterations = getattr(self, 'iterations', 0) + 1
        for n in self.trainable_param_names():
            g = net.grads.get(n)
            if g is None:
                g = np.zeros_like(net.params[n])
            v = self.velocity.get(n)
            if v is None:
                v = np.zeros_like(net.params[n])
            if optimizer == 'adam':
                m = getattr(self, 'moment1', None)
                if m is None:
                    m = self.moment1 = {}
                m1 = m.get(n, np.zeros_like(net.params[n]))
                net.params[n], m[n], self.velocity[n] = O.adam_step(net.params[n], m1, v, g, self.iterations, lr, l2=net.l2[n])
            elif optimizer == 'rmsprop':
                net.params[n], self.velocity[n] = O.rmsprop_step(net.params[n], v, g, lr, l2=net.l2[n])
            else:
                net.params[n], self.velocity[n] = O.sgd_momentum_step(net.params[n], v, g, lr, momentum, net.l2[n])
        for n, val in net.moving_updates.items():
            net.params[n] = val

    def train_step(self, x, labels, dropout_masks=None, lr=1e-2, momentum=0.9):
        total, ce, logits = self.loss_and_grads(x, labels, dropout_masks)
        self.sgd_step(lr, momentum)
        return total, ce
