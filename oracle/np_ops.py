"""CPU oracle: NumPy restatement of the TensorFlow/Keras ops on the DeepLabV3+ hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

PARITY UNPINNED: the reference's arithmetic lives in ``tensorflow==2.11.0``
(/root/reference/requirements.txt:9), which is not installed and cannot be installed here, and the
reference holds no golden vectors.  The TF semantics below are restated from the
published TF/Keras behaviour and triangulated against an independent implementation (torch CPU
ops) in ``tests/test_oracle_ops.py``.

Every op is a forward/backward pair on NHWC arrays.  dtype follows the inputs (float64 is the
authoritative precision, float32 is used for the timed CPU baseline).

Reference call sites restated here:
  * Conv2D / DepthwiseConv2D wrappers      deeplabv3p/models/layers.py:14-31
  * SAME / explicit padding                deeplabv3p/models/layers.py:85-96, deeplabv3p_xception.py:25-54
  * BatchNormalization                     deeplabv3p/models/layers.py:63-70
  * tf.image.resize(bilinear)              deeplabv3p/models/layers.py:48-60
  * AveragePooling2D(pool=(h,w))           deeplabv3p/models/layers.py:132
  * ReLU / ReLU6 / hard-swish              deeplabv3p_mobilenetv2.py:52,61  deeplabv3p_mobilenetv3.py:98-103
  * softmax + sparse CE with ignore index  deeplabv3p/model.py:86, deeplabv3p/loss.py:121-156
  * SGD momentum                           common/model_utils.py:124
"""
import math
import numpy as np

# --------------------------------------------------------------------------------------
# padding
# --------------------------------------------------------------------------------------

def bf16_round(x):
    """round to the nearest bfloat16 (ties to even), returned in x's dtype: the storage rounding of the mixed-precision
    path (a bf16 tensor holds the top 16 bits of the float32 pattern)"""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(np.asarray(x).dtype if np.asarray(x).dtype.kind == 'f' else np.float32)


def same_pad_1d(in_size, k, stride, rate):
    """TF 'SAME' padding for one spatial dim -> (out_size, pad_begin, pad_end).

    out = ceil(in/stride); pad_total = max((out-1)*stride + k_eff - in, 0); the odd unit goes to the
    END (bottom/right).  k_eff = k + (k-1)(rate-1).
    """
    k_eff = k + (k - 1) * (rate - 1)
    out = -(-in_size // stride)
    pad_total = max((out - 1) * stride + k_eff - in_size, 0)
    beg = pad_total // 2
    return out, beg, pad_total - beg


def valid_out_1d(in_size_padded, k, stride, rate):
    k_eff = k + (k - 1) * (rate - 1)
    return (in_size_padded - k_eff) // stride + 1


def resolve_padding(H, W, k, stride, rate, padding):
    """padding: 'same' | 'valid' | (pt, pb, pl, pr) explicit (ZeroPadding2D + VALID).
    returns Ho, Wo, (pt, pb, pl, pr)"""
    if padding == 'same':
        Ho, pt, pb = same_pad_1d(H, k, stride, rate)
        Wo, pl, pr = same_pad_1d(W, k, stride, rate)
        return Ho, Wo, (pt, pb, pl, pr)
    if padding == 'valid':
        padding = (0, 0, 0, 0)
    pt, pb, pl, pr = padding
    return (valid_out_1d(H + pt + pb, k, stride, rate),
            valid_out_1d(W + pl + pr, k, stride, rate), (pt, pb, pl, pr))


def _pad_nhwc(x, pads):
    pt, pb, pl, pr = pads
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def _tap_slice(Ho, Wo, ky, kx, stride, rate):
    ys = slice(ky * rate, ky * rate + (Ho - 1) * stride + 1, stride)
    xs = slice(kx * rate, kx * rate + (Wo - 1) * stride + 1, stride)
    return ys, xs

# --------------------------------------------------------------------------------------
# convolutions (weights in Keras layout: dense HWIO (kh,kw,Cin,Cout); depthwise (kh,kw,C))
# --------------------------------------------------------------------------------------

def conv2d_fwd(x, w, stride=1, rate=1, padding='same', bias=None):
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, Wo, pads = resolve_padding(H, W, kh, stride, rate, padding)
    if kh == 1 and kw == 1 and stride == 1:
        y = (x.reshape(-1, Cin) @ w[0, 0]).reshape(N, H, W, Cout)
    else:
        xp = _pad_nhwc(x, pads)
        y = np.zeros((N, Ho, Wo, Cout), dtype=x.dtype)
        for ky in range(kh):
            for kx in range(kw):
                ys, xs = _tap_slice(Ho, Wo, ky, kx, stride, rate)
                y += (xp[:, ys, xs, :].reshape(-1, Cin) @ w[ky, kx]).reshape(N, Ho, Wo, Cout)
    if bias is not None:
        y = y + bias
    return y


def conv2d_bwd(x, w, gy, stride=1, rate=1, padding='same', need_gx=True):
    """returns (gx, gw, gb)"""
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, Wo, pads = resolve_padding(H, W, kh, stride, rate, padding)
    gy2 = gy.reshape(-1, Cout)
    gb = gy2.sum(0)
    gw = np.zeros_like(w)
    if kh == 1 and kw == 1 and stride == 1:
        gw[0, 0] = x.reshape(-1, Cin).T @ gy2
        gx = (gy2 @ w[0, 0].T).reshape(x.shape) if need_gx else None
        return gx, gw, gb
    xp = _pad_nhwc(x, pads)
    gxp = np.zeros_like(xp) if need_gx else None
    for ky in range(kh):
        for kx in range(kw):
            ys, xs = _tap_slice(Ho, Wo, ky, kx, stride, rate)
            gw[ky, kx] = xp[:, ys, xs, :].reshape(-1, Cin).T @ gy2
            if need_gx:
                gxp[:, ys, xs, :] += (gy2 @ w[ky, kx].T).reshape(N, Ho, Wo, Cin)
    gx = None
    if need_gx:
        pt, pb, pl, pr = pads
        gx = gxp[:, pt:pt + H, pl:pl + W, :]
    return gx, gw, gb


def dwconv2d_fwd(x, w, stride=1, rate=1, padding='same'):
    N, H, W, C = x.shape
    kh, kw, _ = w.shape
    Ho, Wo, pads = resolve_padding(H, W, kh, stride, rate, padding)
    xp = _pad_nhwc(x, pads)
    y = np.zeros((N, Ho, Wo, C), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            ys, xs = _tap_slice(Ho, Wo, ky, kx, stride, rate)
            y += xp[:, ys, xs, :] * w[ky, kx]
    return y


def dwconv2d_bwd(x, w, gy, stride=1, rate=1, padding='same'):
    """returns (gx, gw)"""
    N, H, W, C = x.shape
    kh, kw, _ = w.shape
    Ho, Wo, pads = resolve_padding(H, W, kh, stride, rate, padding)
    xp = _pad_nhwc(x, pads)
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    for ky in range(kh):
        for kx in range(kw):
            ys, xs = _tap_slice(Ho, Wo, ky, kx, stride, rate)
            gw[ky, kx] = np.einsum('nyxc,nyxc->c', xp[:, ys, xs, :], gy)
            gxp[:, ys, xs, :] += gy * w[ky, kx]
    pt, pb, pl, pr = pads
    return gxp[:, pt:pt + H, pl:pl + W, :], gw

# --------------------------------------------------------------------------------------
# batch normalisation (Keras SyncBatchNormalization = non-fused BatchNormalizationBase semantics)
# --------------------------------------------------------------------------------------

def bn_train_fwd(x, gamma, beta, eps):
    """Training-mode BN over (N,H,W).  Normalises with the BIASED batch variance.
    returns y, cache, (batch_mean, biased batch_var); bn_moving_variance_of turns the variance into what the moving
    average takes."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    mean = x2.mean(0)
    var = ((x2 - mean) ** 2).mean(0)
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * invstd
    y = xhat * gamma + beta
    return y, (xhat, invstd, gamma), (mean, var)


def bn_train_bwd(gy, cache):
    """returns gx, ggamma, gbeta"""
    xhat, invstd, gamma = cache
    C = gy.shape[-1]
    g2 = gy.reshape(-1, C)
    xh2 = xhat.reshape(-1, C)
    m = g2.shape[0]
    gbeta = g2.sum(0)
    ggamma = (g2 * xh2).sum(0)
    gx = (gamma * invstd) * (gy - gbeta / m - xhat * (ggamma / m))
    return gx, ggamma, gbeta


def bn_infer_fwd(x, gamma, beta, moving_mean, moving_var, eps):
    scale = gamma / np.sqrt(moving_var + eps)
    return x * scale + (beta - moving_mean * scale)


def bn_moving_variance_of(biased_var, count, rule='biased'):
    """The variance CustomBatchNormalization (layers.py:63-70) feeds into moving_variance.  The layer is
        SyncBatchNormalization  if tf.__version__ >= '2.2'   (layers.py:64-66)
        BatchNormalization      otherwise                    (layers.py:67-68)
    and the test is a STRING compare: True for TF 2.2 .. 2.9, False for the pinned tensorflow==2.11.0
    ('2.11.0' >= '2.2' is False).  The two Keras classes differ in this one statistic:
      'biased'   SyncBatchNormalization refuses fused=True and runs Keras' non-fused BatchNormalizationBase path:
                 `_calculate_mean_and_var` returns E[x^2] - E[x]^2 and that same biased variance is assigned to the
                 moving average;
      'unbiased' plain BatchNormalization on a 4-D input runs the fused FusedBatchNormV3 kernel, whose batch_variance
                 output -- the one Keras' `_fused_batch_norm` feeds to the moving average -- carries Bessel's correction
                 count / (count - 1).
    Both normalise the batch with the biased variance, so training-mode outputs and gradients are identical; only inference
    after training differs (by count / (count - 1) per contributing step: 1 + 1/17423 on a 16 x 33 x 33 map, a factor 2 for
    image_pooling_BN at batch 2).  The product's default is 'biased' (north_star asks for SyncBN over the global batch;
    README.md:38 claims it); get_deeplabv3p_model(..., bn_moving_variance='unbiased') gives the pinned-TF behaviour."""
    if rule == 'unbiased':
        return biased_var * (count / (count - 1.0)) if count > 1 else biased_var
    if rule != 'biased':
        raise ValueError(rule)
    return biased_var


def bn_moving_update(moving, batch_value, momentum):
    return moving * momentum + batch_value * (1.0 - momentum)

# --------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4


def act_fwd(x, act):
    if act == ACT_NONE:
        return x
    if act == ACT_RELU:
        return np.maximum(x, 0)
    if act == ACT_RELU6:
        return np.minimum(np.maximum(x, 0), 6)
    if act == ACT_HSIGMOID:
        return np.minimum(np.maximum(x + 3, 0), 6) * (1.0 / 6.0)
    if act == ACT_HSWISH:
        return x * (np.minimum(np.maximum(x + 3, 0), 6) * (1.0 / 6.0))
    raise ValueError(act)


def act_bwd(x, gy, act):
    """x = pre-activation input.  Sub-gradient conventions follow TF: ReLU'(0)=0, ReLU6'(6)=0."""
    if act == ACT_NONE:
        return gy
    if act == ACT_RELU:
        return gy * (x > 0)
    if act == ACT_RELU6:
        return gy * ((x > 0) & (x < 6))
    if act == ACT_HSIGMOID:
        return gy * (((x + 3) > 0) & ((x + 3) < 6)) * (1.0 / 6.0)
    if act == ACT_HSWISH:
        inner = ((x + 3) > 0) & ((x + 3) < 6)
        hs = np.minimum(np.maximum(x + 3, 0), 6) * (1.0 / 6.0)
        return gy * (hs + x * inner * (1.0 / 6.0))
    raise ValueError(act)

# --------------------------------------------------------------------------------------
# pooling / resize
# --------------------------------------------------------------------------------------

def global_avgpool_fwd(x):
    return x.mean(axis=(1, 2), keepdims=True)


def global_avgpool_bwd(gy, H, W):
    return np.broadcast_to(gy / (H * W), (gy.shape[0], H, W, gy.shape[3])).copy()


def maxpool2d_fwd(x, k, stride, pad):
    """ZeroPadding2D(pad) + MaxPooling2D((k,k), strides, 'valid') (deeplabv3p_resnet50.py:266-267); pad = (top, bottom,
    left, right).  returns (y, argmax) with argmax the flat index of the first maximum inside each window"""
    pt, pb, pl, pr = pad
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    N, Hp, Wp, C = xp.shape
    Ho, Wo = (Hp - k) // stride + 1, (Wp - k) // stride + 1
    win = np.stack([xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :]
                    for ky in range(k) for kx in range(k)], axis=-1)          # (N,Ho,Wo,C,k*k)
    arg = win.argmax(-1)                                                      # first maximum in (ky, kx) order
    return np.take_along_axis(win, arg[..., None], -1)[..., 0], arg


def maxpool2d_bwd(gy, arg, x_shape, k, stride, pad):
    pt, pb, pl, pr = pad
    N, H, W, C = x_shape
    gp = np.zeros((N, H + pt + pb, W + pl + pr, C), dtype=gy.dtype)
    Ho, Wo = gy.shape[1:3]
    for t in range(k * k):
        ky, kx = divmod(t, k)
        sel = np.where(arg == t, gy, 0)
        gp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :] += sel
    return gp[:, pt:pt + H, pl:pl + W, :]                                     # the padding's share is dropped


def bilinear_coeffs(in_size, out_size):
    """tf.image.resize(method='bilinear') in TF2: half_pixel_centers=True, align_corners=False,
    antialias=False.  Source coordinate arithmetic is done in float32 exactly as TF's
    compute_interpolation_weights: scale=in/out (float); src=(o+0.5f)*scale-0.5f;
    lo=max(floor(src),0); hi=min(ceil(src),in-1); lerp=src-floor(src)."""
    scale = np.float32(in_size) / np.float32(out_size)
    o = np.arange(out_size, dtype=np.float32)
    src = (o + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lo = np.maximum(fl.astype(np.int64), 0)
    hi = np.minimum(np.ceil(src).astype(np.int64), in_size - 1)
    t = (src - fl).astype(np.float32)
    return lo, hi, t


def bilinear_matrix(in_size, out_size, dtype):
    lo, hi, t = bilinear_coeffs(in_size, out_size)
    R = np.zeros((out_size, in_size), dtype=dtype)
    r = np.arange(out_size)
    np.add.at(R, (r, lo), (1.0 - t.astype(dtype)))
    np.add.at(R, (r, hi), t.astype(dtype))
    return R


def _apply_axis(R, x, axis):
    xm = np.moveaxis(x, axis, 0)
    shp = xm.shape
    y = (R @ xm.reshape(shp[0], -1)).reshape((R.shape[0],) + shp[1:])
    return np.moveaxis(y, 0, axis)


def resize_bilinear_fwd(x, out_h, out_w):
    N, H, W, C = x.shape
    Ry = bilinear_matrix(H, out_h, x.dtype)
    Rx = bilinear_matrix(W, out_w, x.dtype)
    return _apply_axis(Rx, _apply_axis(Ry, x, 1), 2)


def resize_bilinear_bwd(gy, in_h, in_w):
    N, Ho, Wo, C = gy.shape
    Ry = bilinear_matrix(in_h, Ho, gy.dtype)
    Rx = bilinear_matrix(in_w, Wo, gy.dtype)
    return _apply_axis(Rx.T, _apply_axis(Ry.T, gy, 1), 2)

# --------------------------------------------------------------------------------------
# dropout (mask injected so that stochastic runs are reproducible across implementations)
# --------------------------------------------------------------------------------------

def dropout_fwd(x, keep_mask, rate):
    return x * keep_mask * (1.0 / (1.0 - rate))


def dropout_bwd(gy, keep_mask, rate):
    return gy * keep_mask * (1.0 / (1.0 - rate))

# --------------------------------------------------------------------------------------
# softmax + sparse categorical cross-entropy on probabilities (loss.py:121-156)
# --------------------------------------------------------------------------------------
CE_EPS = 1e-7


def softmax_fwd(z):
    zmax = z.max(-1, keepdims=True)
    e = np.exp(z - zmax)
    return e / e.sum(-1, keepdims=True)


def sparse_ce_fwd_bwd(logits, labels, ignore_index=255):
    """logits (..., C); labels (...) integer-valued float/ints (data.py:116-124).
    returns (loss_mean, probs, dlogits).

    Keras semantics: mask=(y != ignore) [only if ignore_index is truthy, loss.py:139];
    one_hot(255) is an all-zero row; categorical_crossentropy renormalises p/sum(p), clips to
    [1e-7, 1-1e-7] and returns -sum(onehot*log p); Keras reduces with the mean over ALL entries
    (ignored pixels stay in the denominator)."""
    C = logits.shape[-1]
    p = softmax_fwd(logits)
    lab = labels.astype(np.int64)
    flat_p = p.reshape(-1, C)
    flat_l = lab.reshape(-1)
    M = flat_l.shape[0]
    in_range = (flat_l >= 0) & (flat_l < C)
    mask = np.ones(M, dtype=bool)
    if ignore_index:
        mask = flat_l != ignore_index
    active = in_range & mask
    idx = np.where(active)[0]
    pt = flat_p[idx, flat_l[idx]]
    pt_c = np.clip(pt, CE_EPS, 1.0 - CE_EPS)
    loss = -np.log(pt_c).sum() / M
    g = np.zeros_like(flat_p)
    unclipped = (pt > CE_EPS) & (pt < 1.0 - CE_EPS)
    rows = idx[unclipped]
    g[rows] = flat_p[rows]
    g[rows, flat_l[rows]] -= 1.0
    g /= M
    return loss, p, g.reshape(logits.shape)

def _active_rows(logits, labels, ignore_index):
    C = logits.shape[-1]
    flat_l = labels.astype(np.int64).reshape(-1)
    in_range = (flat_l >= 0) & (flat_l < C)
    mask = np.ones(flat_l.shape[0], dtype=bool)
    if ignore_index:
        mask = flat_l != ignore_index
    return flat_l, np.where(in_range & mask)[0]


def weighted_sparse_ce_fwd_bwd(logits, labels, weights, ignore_index=255):
    """WeightedSparseCategoricalCrossEntropy (loss.py:159-191): -w[y] * log(p_y), no clipping, label mask as in the
    plain loss, Keras mean over ALL entries.  returns (loss_mean, probs, dlogits)"""
    C = logits.shape[-1]
    w = np.asarray(weights, dtype=logits.dtype)
    assert w.shape == (C,)
    p = softmax_fwd(logits)
    flat_p = p.reshape(-1, C)
    flat_l, idx = _active_rows(logits, labels, ignore_index)
    M = flat_l.shape[0]
    wy = w[flat_l[idx]]
    loss = -(wy * np.log(flat_p[idx, flat_l[idx]])).sum() / M
    g = np.zeros_like(flat_p)
    g[idx] = flat_p[idx] * wy[:, None]
    g[idx, flat_l[idx]] -= wy
    return loss, p, (g / M).reshape(logits.shape)


def sparse_focal_fwd_bwd(logits, labels, gamma=2.0, alpha=0.25, ignore_index=255):
    """SparseSoftmaxFocalLoss (loss.py:63-118): sum_c alpha * (1 - p_c)^gamma * (-onehot_c * log p_c) with p clipped to
    [1e-15, 1 - 1e-15] = -alpha (1 - p_y)^gamma log p_y; label mask and Keras mean as above.
    d/dz_c = alpha * ((1-p_y)^gamma - gamma (1-p_y)^(gamma-1) p_y log p_y) * (p_c - [c == y])"""
    C = logits.shape[-1]
    p = softmax_fwd(logits)
    flat_p = p.reshape(-1, C)
    flat_l, idx = _active_rows(logits, labels, ignore_index)
    M = flat_l.shape[0]
    pt_raw = flat_p[idx, flat_l[idx]]
    pt = np.clip(pt_raw, 1e-15, 1.0 - 1e-15)
    om = 1.0 - pt
    loss = -(alpha * om ** gamma * np.log(pt)).sum() / M
    f = alpha * (om ** gamma - gamma * om ** (gamma - 1.0) * pt * np.log(pt))
    f = np.where((pt_raw >= 1e-15) & (pt_raw <= 1.0 - 1e-15), f, 0.0)     # the clip has zero slope outside its range
    g = np.zeros_like(flat_p)
    g[idx] = flat_p[idx] * f[:, None]
    g[idx, flat_l[idx]] -= f
    return loss, p, (g / M).reshape(logits.shape)


def loss_fwd_bwd(logits, labels, spec=None, ignore_index=255, sample_weight=None):
    """spec: None / ('ce',) | ('weighted', weights) | ('focal', gamma, alpha)  (train.py:108-137).
    sample_weight (same shape as labels): Keras sample_weight_mode='temporal' (train.py:116-120) -- each pixel's loss is
    multiplied by its weight, the mean still runs over all entries.  Every loss here is a sum of per-pixel terms, so the
    weighted value is assembled from per-pixel evaluations."""
    kind = spec[0] if spec else 'ce'

    def base(z, lab):
        if kind == 'ce':
            return sparse_ce_fwd_bwd(z, lab, ignore_index)
        if kind == 'weighted':
            return weighted_sparse_ce_fwd_bwd(z, lab, spec[1], ignore_index)
        if kind == 'focal':
            return sparse_focal_fwd_bwd(z, lab, spec[1], spec[2], ignore_index)
        raise ValueError(kind)
    if sample_weight is None:
        return base(logits, labels)
    # d(sum_i w_i l_i)/dz_i = w_i d l_i / dz_i: the gradient rows scale by the weights; the value needs the per-pixel
    # losses, obtained by evaluating the (mean-reduced) loss with all other pixels masked out of the gradient path
    loss, p, g = base(logits, labels)
    sw = np.asarray(sample_weight, dtype=logits.dtype).reshape(labels.shape)
    C = logits.shape[-1]
    flat_z, flat_l = logits.reshape(-1, C), labels.reshape(-1)
    M = flat_l.shape[0]
    per = np.array([base(flat_z[i:i + 1], flat_l[i:i + 1])[0] for i in range(M)])       # each is l_i (mean over 1 entry)
    return float((per * sw.reshape(-1)).sum() / M), p, g * sw[..., None]


# --------------------------------------------------------------------------------------
# optimiser: Keras SGD(momentum=0.9, nesterov=False) + l2 regulariser gradient
# --------------------------------------------------------------------------------------
L2_FACTOR = 2e-5  # layers.py:12


def sgd_momentum_step(w, v, g, lr, momentum, l2=0.0):
    """v <- momentum*v - lr*(g + 2*l2*w);  w <- w + v      (common/model_utils.py:124)"""
    g_total = g + (2.0 * l2) * w
    v_new = momentum * v - lr * g_total
    return w + v_new, v_new


def glorot_uniform(rng, shape, fan_in, fan_out, dtype=np.float64):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


def adam_step(w, m, v, g, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, l2=0.0):
    """Keras 2.11 Adam (common/model_utils.py:119): t = 1-based iteration; the l2 regulariser gradient 2*l2*w is part of g"""
    g = g + 2.0 * l2 * w
    m = beta_1 * m + (1.0 - beta_1) * g
    v = beta_2 * v + (1.0 - beta_2) * g * g
    alpha = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    return w - alpha * m / (np.sqrt(v) + epsilon), m, v


def rmsprop_step(w, v, g, lr, rho=0.9, epsilon=1e-7, l2=0.0):
    """Keras 2.11 RMSprop(momentum=0, centered=False) (common/model_utils.py:121): w -= lr * g * rsqrt(v + eps)"""
    g = g + 2.0 * l2 * w
    v = rho * v + (1.0 - rho) * g * g
    return w - lr * g / np.sqrt(v + epsilon), v


def he_normal(rng, shape, fan_in, dtype=np.float64):
    """Keras he_normal = VarianceScaling(2, 'fan_in', 'truncated_normal') (deeplabv3p_resnet50.py: every conv)"""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(dtype)


# ------------------------------------------------------------------- label tail of the generator (deeplabv3p/data.py)
def prepare_labels(label, num_classes, ignore_index=255, adaptive=False):
    """deeplabv3p/data.py:116-145 for one image: label (P,) integers -> (float32 labels with every value above
    num_classes-1 set to ignore_index, float32 pixel weights or None).  The weights restate sklearn's
    compute_class_weight('balanced', classes=np.unique(label), y=label): len(y) / (n_distinct * bincount), float64,
    written into the generator's float32 array (ignore_index counts as a value, as it does there)"""
    lab = np.asarray(label).astype(np.int32).ravel().copy()
    lab[lab > num_classes - 1] = ignore_index
    if not adaptive:
        return lab.astype(np.float32), None
    values, inverse, counts = np.unique(lab, return_inverse=True, return_counts=True)
    recip = lab.size / (len(values) * counts.astype(np.float64))
    return lab.astype(np.float32), recip[inverse].astype(np.float32)


# ---------------------------------------------------------------------------------------- evaluation (eval.py)
def confusion_matrix(gt_mask, pred_mask, num_classes):
    """eval.py:368-373 generate_matrix: rows = ground truth, columns = prediction, labels outside [0, C) dropped"""
    gt = np.asarray(gt_mask).astype(np.int64).ravel()
    pr = np.asarray(pred_mask).astype(np.int64).ravel()
    valid = (gt >= 0) & (gt < num_classes)
    label = num_classes * gt[valid] + pr[valid]
    return np.bincount(label, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def jaccard_metric(labels, pred, num_classes):
    """deeplabv3p/metrics.py:29-46 Jaccard: labels, pred (N, P) integer arrays.  For every class i in [0, C]: per-image
    IoU over the images that contain the class, averaged; then the mean over the classes that occur at all"""
    labels = np.asarray(labels).astype(np.int64)
    pred = np.asarray(pred).astype(np.int64)
    iou = []
    for i in range(num_classes + 1):
        t, q = labels == i, pred == i
        inter, union = (t & q).sum(1), (t | q).sum(1)
        legal = t.sum(1) > 0
        if legal.any():
            iou.append(float(np.mean(inter[legal] / union[legal])))
    return float(np.mean(iou)) if iou else float('nan')


def miou_summary(cm):
    """eval.py:462-497: pixel accuracy, per-class accuracy / IoU / Dice / frequency, mean IoU (NaN -> 0 before the
    mean, as the reference does), frequency-weighted IoU"""
    cm = np.asarray(cm, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        pixel_acc = np.diag(cm).sum() / cm.sum()
        class_acc = np.diag(cm) / cm.sum(axis=1)
        class_acc[np.isnan(class_acc)] = 0
        inter = np.diag(cm)
        union = cm.sum(axis=0) + cm.sum(axis=1) - inter
        iou = inter / union
        iou[np.isnan(iou)] = 0
        freq = cm.sum(axis=1) / cm.sum()
        freq[np.isnan(freq)] = 0
        dice = 2 * inter / (union + inter)
        dice[np.isnan(dice)] = 0
    return {'PixelAcc': float(pixel_acc), 'ClassAcc': class_acc, 'mClassAcc': float(np.nanmean(class_acc)), 'IoU': iou,
            'mIoU': float(np.nanmean(iou)), 'Freq': freq, 'FWIoU': float((freq[freq > 0] * iou[freq > 0]).sum()),
            'Dice': dice}

