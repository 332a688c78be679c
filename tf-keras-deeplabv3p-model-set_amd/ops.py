"""Tensor-level wrappers over the C ABI (include/dl3p.h).

PyTorch is only the allocator / stream provider here: every function takes CUDA(=HIP) float32
tensors in NHWC memory (a channel slice of a wider buffer is fine: row stride `ld` is taken from the
tensor's strides), hands raw device pointers to libdl3p.so and returns.  There is no CPU or torch
fallback: a missing library raises at import, a non-HIP tensor raises here.
"""
import ctypes
import torch

from ._lib import lib, Dl3pError  # noqa: F401

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4
MAX_STAT_ROWS = 2048


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pl(t):
    """(device pointer, row stride in floats) of an NHWC / [M,C] tensor or channel-slice view"""
    if t is None:
        return None, 0
    if not t.is_cuda or t.dtype != torch.float32:
        raise Dl3pError('dl3p ops need float32 tensors on the HIP device (got %s on %s)' % (t.dtype, t.device))
    if t.stride(-1) != 1:
        raise Dl3pError('channel dim must be contiguous')
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    # rows must be evenly spaced: stride of each leading dim == size*stride of the next
    for d in range(t.dim() - 2):
        if t.shape[d + 1] > 0 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
            raise Dl3pError('rows are not evenly strided: %s %s' % (tuple(t.shape), t.stride()))
    return t.data_ptr(), ld


def _p(t):
    return None if t is None else t.data_ptr()


def same_pad(in_size, k, stride, rate):
    """TF SAME: (out, pad_begin) -- the odd unit of padding goes to the end"""
    k_eff = k + (k - 1) * (rate - 1)
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return out, total // 2


def conv_geometry(H, W, k, stride, rate, padding):
    """padding 'same' | 'valid' | (pt, pb, pl, pr) -> Ho, Wo, pad_t, pad_l"""
    if padding == 'same':
        Ho, pt = same_pad(H, k, stride, rate)
        Wo, pl = same_pad(W, k, stride, rate)
        return Ho, Wo, pt, pl
    if padding == 'valid':
        padding = (0, 0, 0, 0)
    pt, pb, pl, pr = padding
    k_eff = k + (k - 1) * (rate - 1)
    return (H + pt + pb - k_eff) // stride + 1, (W + pl + pr - k_eff) // stride + 1, pt, pl


def new_partials(C, device, rows=MAX_STAT_ROWS):
    return torch.empty(rows * 2 * C, dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------------- depthwise
def _arm_upsampled(up):
    """up (N,h,w,C1): the NEXT depthwise launch forms channels [0, C1) of its input from it (dl3p_dw_upsampled_input)"""
    if up is not None:
        upp, upld = _pl(up)
        lib().dw_upsampled_input(upp, upld, up.shape[1], up.shape[2], up.shape[3], _stream())


def dwconv2d_fwd(x, w, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE,
                 out=None, partials=None, upsampled=None):
    """x (N,H,W,C); w (k,k,C) or (k,k,C,1) -> y (N,Ho,Wo,C) [, rows]"""
    _arm_upsampled(upsampled)
    N, H, W, C = x.shape
    k = w.shape[0]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    y = out if out is not None else torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().dwconv2d_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w), yp, ldy, _p(partials), ctypes.byref(rows),
                       N, H, W, C, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def dwconv2d_bwd_data(dy, w, x_shape, stride=1, rate=1, padding='same', out=None, accumulate=False):
    N, H, W, C = x_shape
    k = w.shape[0]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().dwconv2d_bwd_data(dp, ldd, _p(w), gp, ldg, int(accumulate), N, H, W, C, k, stride, rate, pt, pl, Ho, Wo,
                            _stream())
    return gx


def dwconv2d_bwd_data_bn(dy, w, x_shape, z, scale, shift, act, mean, invstd, partials, stride=1, rate=1, padding='same',
                          out=None, accumulate=False):
    """dwconv2d_bwd_data + the BatchNorm-backward partial sums of act(BN(z)) of the finished gradient; -> (gx, rows)"""
    N, H, W, C = x_shape
    k = w.shape[0]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    dp, lddy = _pl(dy)
    gp, ldg = _pl(gx)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    lib().dwconv2d_bwd_data_bn(dp, lddy, _p(w), gp, ldg, int(accumulate), N, H, W, C, k, stride, rate, pt, pl, Ho, Wo,
                               zp, ldz, _p(scale), _p(shift), act, _p(mean), _p(invstd), _p(partials),
                               ctypes.byref(rows), _stream())
    return gx, rows.value


def dwconv2d_bwd_weight(x, dy, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE,
                        workspace=None, upsampled=None):
    _arm_upsampled(upsampled)
    N, H, W, C = x.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    need = lib().dwconv2d_bwd_weight_workspace(N, Ho, Wo, C, k)
    ws = workspace if workspace is not None else torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((k, k, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    lib().dwconv2d_bwd_weight(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(gw), _p(ws), ws.numel() * 4,
                              N, H, W, C, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return gw


def dwconv2d_bwd_weight_bn(x, g, z, bn, act, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None,
                           in_act=ACT_NONE, want_dz=True, upsampled=None):
    """depthwise weight gradient with the BatchNorm-backward apply of the conv's output folded in
    (dl3p_dwconv2d_bwd_weight_slabs_bn + the slab reduction) -> (gw, dz)"""
    _arm_upsampled(upsampled)
    N, H, W, C = x.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    need = lib().dwconv2d_bwd_weight_workspace(N, Ho, Wo, C, k)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((k, k, C), dtype=torch.float32, device=x.device)
    dz = torch.empty(g.shape, dtype=torch.float32, device=x.device) if want_dz else None
    xp, ldx = _pl(x)
    gp, ldg = _pl(g)
    zp, ldz = _pl(z)
    dp, ldd = _pl(dz)
    rows = ctypes.c_int(0)
    lib().dwconv2d_bwd_weight_slabs_bn(xp, ldx, _p(in_scale), _p(in_shift), in_act, gp, ldg, zp, ldz, _p(bn.scale),
                                       _p(bn.shift), act, _p(bn.mean), _p(bn.invstd), _p(bn.coef), dp, ldd or 0, _p(ws),
                                       ws.numel() * 4, ctypes.byref(rows), N, H, W, C, k, stride, rate, pt, pl, Ho, Wo,
                                       _stream())
    lib().reduce_rows(_p(ws), rows.value, k * k * C, _p(gw), 0, _stream())
    return gw, dz


# ------------------------------------------------------------------------------------- pointwise
def _rows(t):
    return t.numel() // t.shape[-1]


def pwconv_fwd(x, w, bias=None, in_scale=None, in_shift=None, in_act=ACT_NONE, out=None, partials=None):
    """x (..., K); w (K,N) -> y (..., N)"""
    K, Nn = w.shape[-2], w.shape[-1]
    M = _rows(x)
    y = out if out is not None else torch.empty(tuple(x.shape[:-1]) + (Nn,), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().pwconv_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w), _p(bias), yp, ldy, _p(partials),
                     ctypes.byref(rows), M, K, Nn, _stream())
    return (y, rows.value) if partials is not None else y


def pwconv_fwd_wt(x, wt, bias=None, in_scale=None, in_shift=None, in_act=ACT_NONE, out=None, partials=None):
    """pwconv_fwd with the kernel given transposed, wt (N, K)"""
    M, K = _rows(x), x.shape[-1]
    N = wt.shape[0]
    y = out if out is not None else torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().pwconv_fwd_wt(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(wt), _p(bias), yp, ldy, _p(partials),
                        ctypes.byref(rows), M, K, N, _stream())
    return (y, rows.value) if partials is not None else y


def pwconv_fwd_wt_splitk(x, wt, bias=None, in_scale=None, in_shift=None, in_act=ACT_NONE, out=None, partials=None):
    """pwconv_fwd_wt through the split-K form (few rows, long reduction); raises Dl3pError where dl3p_pwconv_fwd_splitk_plan says 0"""
    M, K = _rows(x), x.shape[-1]
    N = wt.shape[0]
    y = out if out is not None else torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    wsb = lib().pwconv_fwd_splitk_workspace(M, K, N)
    ws = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device=x.device)
    lib().pwconv_fwd_wt_splitk(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(wt), _p(bias), yp, ldy, _p(partials),
                               ctypes.byref(rows), _p(ws), wsb, M, K, N, _stream())
    return (y, rows.value) if partials is not None else y


def transpose_batch(src, dst, table):
    lib().transpose_batch(_p(src), _p(dst), _p(table), int(table.shape[0]), _stream())


def pwconv_bwd_data(dy, w, out=None, accumulate=False):
    K, Nn = w.shape[-2], w.shape[-1]
    M = _rows(dy)
    gx = out if out is not None else torch.empty(tuple(dy.shape[:-1]) + (K,), dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().pwconv_bwd_data(dp, ldd, _p(w), gp, ldg, int(accumulate), M, K, Nn, _stream())
    return gx


def pwconv_bwd_data_bn(dy, w, z, scale, shift, act, mean, invstd, partials, out=None, accumulate=False):
    """pwconv_bwd_data + the BatchNorm-backward partial sums of act(BN(z)) from the finished gradient; -> (gx, rows)"""
    M, N = _rows(dy), dy.shape[-1]
    K = w.shape[0]
    gx = out if out is not None else torch.empty(dy.shape[:-1] + (K,), dtype=torch.float32, device=dy.device)
    dp, lddy = _pl(dy)
    gp, ldg = _pl(gx)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    lib().pwconv_bwd_data_bn(dp, lddy, _p(w), gp, ldg, int(accumulate), M, K, N, zp, ldz, _p(scale), _p(shift), act,
                             _p(mean), _p(invstd), _p(partials), ctypes.byref(rows), _stream())
    return gx, rows.value


def pwconv_bwd_weight(x, dy, in_scale=None, in_shift=None, in_act=ACT_NONE, with_bias=False, workspace=None):
    K, Nn = x.shape[-1], dy.shape[-1]
    M = _rows(x)
    need = lib().pwconv_bwd_weight_workspace(M, K, Nn)
    ws = workspace if workspace is not None else torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((K, Nn), dtype=torch.float32, device=x.device)
    gb = torch.empty((Nn,), dtype=torch.float32, device=x.device) if with_bias else None
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    lib().pwconv_bwd_weight(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(gw), _p(gb), _p(ws),
                            ws.numel() * 4, M, K, Nn, _stream())
    return (gw, gb) if with_bias else gw


def pwconv_bwd_weight_bn(x, g, z, bn, act, in_scale=None, in_shift=None, in_act=ACT_NONE, want_dz=True):
    """weight gradient with the BatchNorm-backward apply of the conv's output folded in (dl3p_pwconv_bwd_weight_slabs_bn
    + the slab reduction): g = gradient of act(bn(z)), bn.coef from bn_bwd_finalize -> (gw, dz)"""
    K, Nn = x.shape[-1], g.shape[-1]
    M = _rows(x)
    need = lib().pwconv_bwd_weight_workspace(M, K, Nn)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((K, Nn), dtype=torch.float32, device=x.device)
    dz = torch.empty(g.shape, dtype=torch.float32, device=x.device) if want_dz else None
    xp, ldx = _pl(x)
    gp, ldg = _pl(g)
    zp, ldz = _pl(z)
    dp, ldd = _pl(dz)
    rows = ctypes.c_int(0)
    lib().pwconv_bwd_weight_slabs_bn(xp, ldx, _p(in_scale), _p(in_shift), in_act, gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift),
                                     act, _p(bn.mean), _p(bn.invstd), _p(bn.coef), dp, ldd or 0, _p(ws), ws.numel() * 4,
                                     ctypes.byref(rows), M, K, Nn, _stream())
    lib().reduce_rows(_p(ws), rows.value, K * Nn, _p(gw), 0, _stream())
    return gw, dz


# ------------------------------------------------------------------------------------- dense conv
def conv2d_fwd(x, w, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE, out=None,
               partials=None):
    N, H, W, Cin = x.shape
    k, _, _, Cout = w.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    y = out if out is not None else torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().conv2d_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w), yp, ldy, _p(partials), ctypes.byref(rows),
                     N, H, W, Cin, Cout, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def conv2d_bwd_weight(x, dy, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE):
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    need = lib().conv2d_bwd_weight_workspace(N, Ho, Wo, Cin, Cout, k)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((k, k, Cin, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    lib().conv2d_bwd_weight(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(gw), _p(ws), ws.numel() * 4,
                            N, H, W, Cin, Cout, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return gw


def conv2d_bwd_data(dy, w, x_shape, stride=1, rate=1, padding='same', out=None, accumulate=False):
    N, H, W, Cin = x_shape
    k, _, _, Cout = w.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().conv2d_bwd_data(dp, ldd, _p(w), gp, ldg, int(accumulate), N, H, W, Cin, Cout, k, stride, rate, pt, pl,
                          Ho, Wo, _stream())
    return gx


# ------------------------------------------------------------------------------------- batch norm
class BNState:
    """device-side state of one BatchNormalization layer (all float32 [C])"""

    def __init__(self, C, device, eps=1e-3, momentum=0.99):
        f = dict(dtype=torch.float32, device=device)
        self.C, self.eps, self.momentum = C, eps, momentum
        self.gamma = torch.ones(C, **f)
        self.beta = torch.zeros(C, **f)
        self.moving_mean = torch.zeros(C, **f)
        self.moving_var = torch.ones(C, **f)
        self.scale = torch.empty(C, **f)
        self.shift = torch.empty(C, **f)
        self.mean = torch.empty(C, **f)
        self.invstd = torch.empty(C, **f)
        self.coef = torch.empty(3 * C, **f)
        self.dgamma = torch.empty(C, **f)
        self.dbeta = torch.empty(C, **f)


def bn_finalize(bn, partials, rows, count, update_moving=True, sums=None):
    lib().bn_finalize(_p(partials), rows, _p(sums), bn.C, float(count), _p(bn.gamma), _p(bn.beta), bn.eps,
                      bn.momentum, _p(bn.moving_mean), _p(bn.moving_var), int(update_moving), _p(bn.scale),
                      _p(bn.shift), _p(bn.mean), _p(bn.invstd), _stream())


def bn_reduce_partials(partials, rows, C2):
    sums = torch.empty(C2, dtype=torch.float64, device=partials.device)
    lib().bn_reduce_partials(_p(partials), rows, C2, _p(sums), _stream())
    return sums


def bn_infer_coeffs(bn):
    lib().bn_infer_coeffs(_p(bn.gamma), _p(bn.beta), _p(bn.moving_mean), _p(bn.moving_var), bn.eps, _p(bn.scale),
                          _p(bn.shift), _p(bn.mean), _p(bn.invstd), bn.C, _stream())


def bn_backward(bn, g, z, act, partials, frozen=False, out=None, sums=None):
    """g: gradient w.r.t. act(bn(z)); returns dz (in place in g unless out is given)"""
    M = _rows(z)
    gp, ldg = _pl(g)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    if not frozen and sums is None:
        lib().bn_bwd_reduce(gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift), act, _p(bn.mean), _p(bn.invstd),
                            _p(partials), ctypes.byref(rows), M, bn.C, _stream())
    lib().bn_bwd_finalize(_p(partials), rows.value, _p(sums), bn.C, float(M), _p(bn.gamma), _p(bn.invstd),
                          _p(bn.scale), int(frozen), _p(bn.dgamma), _p(bn.dbeta), _p(bn.coef), _stream())
    dz = out if out is not None else g
    dp, ldd = _pl(dz)
    lib().bn_bwd_apply(gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift), act, _p(bn.mean), _p(bn.invstd), _p(bn.coef),
                       dp, ldd, 0, M, bn.C, _stream())
    return dz


# ------------------------------------------------------------------------------------- elementwise
def affine_act(x, scale=None, shift=None, act=ACT_NONE, residual=None, rscale=None, rshift=None, ract=ACT_NONE,
               dropout_rate=0.0, seed=0, step_counter=None, out=None):
    C = x.shape[-1]
    y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    rp, ldr = _pl(residual)
    yp, ldy = _pl(y)
    lib().affine_act(xp, ldx, _p(scale), _p(shift), act, rp, ldr, _p(rscale), _p(rshift), ract, float(dropout_rate),
                     int(seed), _p(step_counter), yp, ldy, _rows(x), C, _stream())
    return y


def scale_mask_bwd(gy, dropout_rate=0.0, seed=0, step_counter=None, out=None, accumulate=False):
    C = gy.shape[-1]
    gx = out if out is not None else torch.empty(gy.shape, dtype=torch.float32, device=gy.device)
    gp, ldg = _pl(gy)
    xp, ldx = _pl(gx)
    lib().scale_mask_bwd(gp, ldg, float(dropout_rate), int(seed), _p(step_counter), xp, ldx, int(accumulate),
                         _rows(gy), C, _stream())
    return gx


def dropout_mask(shape, rate, seed, step_counter, device):
    C = shape[-1]
    m = torch.empty(shape, dtype=torch.float32, device=device)
    lib().dropout_mask(float(rate), int(seed), _p(step_counter), _p(m), m.numel() // C, C, _stream())
    return m


def _pool_ws(N, HW, C, device, chunked):
    """zeroed ticket + partial-row workspace of the chunked per-image reductions (None: one workgroup per image)"""
    if not chunked:
        return None, 0
    nbytes = lib().pool_workspace(N, HW, C)
    return torch.zeros(nbytes // 4, dtype=torch.float32, device=device), nbytes


def global_avgpool_fwd(x, in_scale=None, in_shift=None, in_act=ACT_NONE, out_scale=1.0, out=None, chunked=True):
    N, H, W, C = x.shape
    y = out if out is not None else torch.empty((N, 1, 1, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    ws, wsb = _pool_ws(N, H * W, C, x.device, chunked)
    lib().global_avgpool_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, yp, ldy, float(out_scale), N, H * W, C,
                             _p(ws), wsb, _stream())
    return y


def global_avgpool_bwd(gy, H, W, out=None, accumulate=False):
    N, C = gy.shape[0], gy.shape[-1]
    gx = out if out is not None else torch.empty((N, H, W, C), dtype=torch.float32, device=gy.device)
    gp, ldg = _pl(gy)
    xp, ldx = _pl(gx)
    lib().global_avgpool_bwd(gp, ldg, xp, ldx, int(accumulate), N, H * W, C, _stream())
    return gx


def maxpool2d_fwd(x, k, stride, pad, in_scale=None, in_shift=None, in_act=ACT_NONE, argmax=None):
    """x (N,H,W,C), explicit zero padding pad = (top, bottom, left, right) then VALID max pooling"""
    N, H, W, C = x.shape
    pt, pb, pl, pr = pad
    Ho, Wo = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    lib().maxpool2d_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, yp, ldy, _p(argmax), N, H, W, C, k, stride, pt, pl, Ho, Wo,
                        _stream())
    return y


def maxpool2d_bwd(x, dy, k, stride, pad, in_scale=None, in_shift=None, in_act=ACT_NONE, out=None, accumulate=False,
                  argmax=None):
    N, H, W, C = x.shape
    pt, pb, pl, pr = pad
    Ho, Wo = dy.shape[1], dy.shape[2]
    gx = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().maxpool2d_bwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(argmax), gp, ldg, int(accumulate), N, H, W, C,
                        k, stride, pt, pl, Ho, Wo, _stream())
    return gx


def resize_bilinear_fwd(x, H, W, out=None):
    N, h, w, C = x.shape
    y = out if out is not None else torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    lib().resize_bilinear_fwd(xp, ldx, yp, ldy, N, h, w, C, H, W, _stream())
    return y


def resize_bilinear_bwd(gy, h, w, out=None, accumulate=False):
    N, H, W, C = gy.shape
    gx = out if out is not None else torch.empty((N, h, w, C), dtype=torch.float32, device=gy.device)
    gp, ldg = _pl(gy)
    xp, ldx = _pl(gx)
    lib().resize_bilinear_bwd(gp, ldg, xp, ldx, int(accumulate), N, h, w, C, H, W, _stream())
    return gx


# ------------------------------------------------------------------------------------- head / loss
def upsample_softmax_ce(z, C, H, W, labels=None, ignore_index=255, want_probs=False, want_logits=False,
                        want_grad=False, ld_big=None, loss=('ce',), pixel_weights=None):
    """z (N,h,w,Cpad) small logits -> dict(loss, probs (N,H,W,C), logits (N,H,W,ld_big), dlogits)"""
    N, h, w, _ = z.shape
    dev = z.device
    ld_big = ld_big or ((C + 3) // 4) * 4
    zp, ldz = _pl(z)
    out = {}
    probs = torch.empty((N, H, W, C), dtype=torch.float32, device=dev) if want_probs else None
    logits = torch.zeros((N, H, W, ld_big), dtype=torch.float32, device=dev) if want_logits else None
    dlog = torch.zeros((N, H, W, ld_big), dtype=torch.float32, device=dev) if want_grad else None
    partial = torch.zeros(MAX_STAT_ROWS, dtype=torch.float32, device=dev) if labels is not None else None
    rows = ctypes.c_int(0)
    inv = 1.0 / float(N * H * W)
    # loss: ('ce',) | ('weighted', weight tensor [C]) | ('focal', gamma, alpha)
    kind = {'ce': 0, 'weighted': 1, 'focal': 2}[loss[0]]
    cw = loss[1] if kind == 1 else None
    gamma, alpha = (float(loss[1]), float(loss[2])) if kind == 2 else (0.0, 0.0)
    lib().upsample_softmax_loss(zp, ldz, _p(labels), int(ignore_index or 0), inv, kind, _p(cw), gamma, alpha,
                                _p(pixel_weights), _p(logits), _p(probs), _p(dlog), ld_big, _p(partial), ctypes.byref(rows),
                                N, h, w, C, H, W, _stream())
    if labels is not None:
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        lib().reduce_rows(_p(partial), rows.value, 1, _p(loss), 0, _stream())
        out['loss'] = loss
    out['probs'], out['logits'], out['dlogits'] = probs, logits, dlog
    return out


def argmax_confusion(z, C, H, W, labels=None, confusion=None, want_mask=False):
    """z (N,h,w,Cpad) small logits -> (pred mask (N,H,W) int32 or None, confusion (C,C) int64 or None); an existing
    `confusion` tensor is accumulated into (eval.py:443)"""
    N, h, w, _ = z.shape
    zp, ldz = _pl(z)
    pred = torch.empty((N, H, W), dtype=torch.int32, device=z.device) if want_mask else None
    if labels is not None and confusion is None:
        confusion = torch.zeros((C, C), dtype=torch.int64, device=z.device)
    lib().argmax_confusion(zp, ldz, _p(labels), _p(pred), _p(confusion) if labels is not None else None, N, h, w, C, H, W,
                           _stream())
    return pred, (confusion if labels is not None else None)


def label_prepare(labels_u8, num_classes, ignore_index=255, adaptive=False, labels_out=None, weights_out=None):
    """uint8 labels (N, P) on the device -> (float32 labels with values above num_classes-1 set to ignore_index,
    float32 'balanced' pixel weights or None): deeplabv3p/data.py:116-145"""
    N, P = labels_u8.shape
    assert labels_u8.dtype == torch.uint8 and labels_u8.is_contiguous()
    if labels_out is None:
        labels_out = torch.empty((N, P), dtype=torch.float32, device=labels_u8.device)
    hist = None
    if adaptive:
        if weights_out is None:
            weights_out = torch.empty((N, P), dtype=torch.float32, device=labels_u8.device)
        hist = torch.empty(N * 256, dtype=torch.int32, device=labels_u8.device)
    lib().label_prepare(labels_u8.data_ptr(), labels_out.data_ptr(), _p(weights_out) if adaptive else None, _p(hist), N, P,
                        num_classes, ignore_index, _stream())
    return labels_out, (weights_out if adaptive else None)


def head_train_supported(h, w, C, H, W):
    return bool(lib().head_train_supported(h, w, C, H, W))


def head_train_rows_supported(h, w, C, H, W):
    return bool(lib().head_train_rows_supported(h, w, C, H, W))


def head_train(z, C, H, W, labels, ignore_index=255, rows_form=False):
    """fused training head: z (N,h,w,Cpad) -> (loss [1], d loss / d z (N,h,w,Cpad)) without the (N,H,W,C) gradient;
    rows_form: dl3p_head_train_rows (the row-walking kernel) instead of dl3p_head_train (the tile kernel)"""
    N, h, w, cp = z.shape
    dev = z.device
    zp, ldz = _pl(z)
    gz = torch.zeros((N, h, w, cp), dtype=torch.float32, device=dev)
    partial = torch.zeros(MAX_STAT_ROWS, dtype=torch.float32, device=dev)
    rows = ctypes.c_int(0)
    if rows_form:
        wsb = lib().head_train_rows_workspace(N, h, w, C, H, W)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
        lib().head_train_rows(zp, ldz, _p(labels), int(ignore_index or 0), 1.0 / float(N * H * W), _p(gz), cp, 0, _p(partial),
                              ctypes.byref(rows), _p(ws), wsb, N, h, w, C, H, W, _stream())
    else:
        lib().head_train(zp, ldz, _p(labels), int(ignore_index or 0), 1.0 / float(N * H * W), _p(gz), cp, 0, _p(partial),
                         ctypes.byref(rows), N, h, w, C, H, W, _stream())
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    lib().reduce_rows(_p(partial), rows.value, 1, _p(loss), 0, _stream())
    return loss, gz


def sgd_momentum(w, v, g, lr_dev, momentum=0.9, l2=0.0, grad_scale=1.0, l2_elem=None, lr_scale_elem=None):
    lib().sgd_momentum(_p(w), _p(v), _p(g), w.numel(), _p(lr_dev), float(momentum), float(l2), float(grad_scale),
                       _p(l2_elem), _p(lr_scale_elem), _stream())


def act_bwd(g, z, act, out, accumulate=False):
    """out (+)= g * act'(z): backward of a bare activation on a materialised tensor"""
    gp, ldg = _pl(g)
    zp, ldz = _pl(z)
    op, ldo = _pl(out)
    lib().bn_bwd_apply(gp, ldg, zp, ldz, None, None, act, None, None, None, op, ldo, int(accumulate), _rows(z),
                       z.shape[-1], _stream())
    return out


def im2col(x, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE):
    N, H, W, Cin = x.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    kp = (k * k * Cin + 3) // 4 * 4
    col = torch.empty((N, Ho, Wo, kp), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    lib().im2col(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(col), kp, N, H, W, Cin, k, stride, rate, pt, pl, Ho, Wo,
                 _stream())
    return col


def conv2d_gemm_supported(Cin, Cout, k, stride):
    return bool(lib().conv2d_gemm_supported(Cin, Cout, k, stride))


def conv2d_gemm_fwd(x, w, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE, bias=None,
                    partials=None):
    """dense conv as an implicit GEMM (no im2col matrix): x (N,H,W,Cin), w (k,k,Cin,Cout) -> y (N,Ho,Wo,Cout) [, rows]"""
    N, H, W, Cin = x.shape
    k, Cout = w.shape[0], w.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    wt = w.reshape(k * k * Cin, Cout).t().contiguous()
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    rows = ctypes.c_int(0)
    lib().conv2d_gemm_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(wt), _p(bias), _p(y), Cout, _p(partials),
                          ctypes.byref(rows), N, H, W, Cin, Cout, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def conv2d_gemm_bwd_data(dy, w, x_shape, stride=1, rate=1, padding='same', out=None, accumulate=False):
    N, H, W, Cin = x_shape
    k, Cout = w.shape[0], w.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    wd = torch.empty(Cin * k * k * Cout, dtype=torch.float32, device=dy.device)
    lib().conv2d_gemm_dgrad_weights(_p(w.contiguous()), _p(wd), k, Cin, Cout, _stream())
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().conv2d_gemm_bwd_data(dp, ldd, _p(wd), gp, ldg, int(accumulate), N, H, W, Cin, Cout, k, stride, rate, pt, pl, Ho,
                               Wo, _stream())
    return gx


def conv2d_gemm_bwd_weight(x, dy, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE,
                           with_bias=False):
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    need = lib().conv2d_gemm_bwd_weight_workspace(N, Ho, Wo, Cin, Cout, k)
    ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((k, k, Cin, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    gb = torch.empty(Cout, dtype=torch.float32, device=x.device) if with_bias else None
    lib().conv2d_gemm_bwd_weight(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(gw), _p(gb), _p(ws), need, N, H, W,
                                 Cin, Cout, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (gw, gb) if with_bias else gw


def conv2d_gemm_fwd_sb(x, w, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE, bias=None,
                       partials=None):
    """conv2d_gemm_fwd on the split-bf16 kernel (the kernel operand pre-split here: [3][Cout][pitch >= k k Cin])"""
    N, H, W, Cin = x.shape
    k, Cout = w.shape[0], w.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    wsp = split_bf16x3(w.reshape(k * k * Cin, Cout).t().contiguous())
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    rows = ctypes.c_int(0)
    lib().conv2d_gemm_fwd_sb(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(wsp), wsp.shape[2], _p(bias), _p(y), Cout, _p(partials),
                             ctypes.byref(rows), N, H, W, Cin, Cout, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def conv2d_gemm_bwd_data_sb(dy, w, x_shape, stride=1, rate=1, padding='same', out=None, accumulate=False):
    """conv2d_gemm_bwd_data on the split-bf16 kernel (wd = conv2d_gemm_dgrad_weights(w) pre-split: [3][Cin][pitch >= k k Cout])"""
    N, H, W, Cin = x_shape
    k, Cout = w.shape[0], w.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    wd = torch.empty((Cin, k * k * Cout), dtype=torch.float32, device=dy.device)
    lib().conv2d_gemm_dgrad_weights(_p(w.contiguous()), _p(wd), k, Cin, Cout, _stream())
    wdsp = split_bf16x3(wd)
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    lib().conv2d_gemm_bwd_data_sb(dp, ldd, _p(wdsp), wdsp.shape[2], gp, ldg, int(accumulate), N, H, W, Cin, Cout, k, stride, rate,
                                  pt, pl, Ho, Wo, _stream())
    return gx


def stem_conv_supported(Cin, Cout, k, stride, rate):
    return bool(lib().stem_conv_supported(Cin, Cout, k, stride, rate))


def stem_conv_fwd(x, w, padding='same', partials=None):
    """the RGB stem (3x3 stride 2) without an im2col matrix: x (N,H,W,3); w (3,3,3,Cout) -> y (N,Ho,Wo,Cout) [, rows]"""
    N, H, W, Cin = x.shape
    Cout = w.shape[-1]
    assert stem_conv_supported(Cin, Cout, w.shape[0], 2, 1)
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, 2, 1, padding)
    wk = torch.zeros((28, Cout), dtype=torch.float32, device=x.device)
    wk[:27] = w.reshape(27, Cout)
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    rows = ctypes.c_int(0)
    lib().stem_conv_fwd(xp, ldx, _p(wk), _p(y), Cout, _p(partials), ctypes.byref(rows), N, H, W, Cout, pt, pl, Ho, Wo,
                        _stream())
    return (y, rows.value) if partials is not None else y


def stem_conv_bwd_weight(x, dy, padding='same'):
    """-> gw (3,3,3,Cout)"""
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, 2, 1, padding)
    need = lib().stem_conv_bwd_weight_workspace(N, Ho, Wo, Cout)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.full((28, Cout), float('nan'), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    lib().stem_conv_bwd_weight(xp, ldx, dp, ldd, _p(gw), _p(ws), need, N, H, W, Cout, pt, pl, Ho, Wo, _stream())
    assert float(gw[27].abs().max()) == 0.0
    return gw[:27].reshape(3, 3, 3, Cout)


def stem_conv_bwd_weight_bn(x, g, z, bn, act, padding='same'):
    """stem weight gradient with the BatchNorm-backward apply of (g, z) folded in (bn: BNState after bn_backward's finalize) -> gw"""
    N, H, W, Cin = x.shape
    Cout = g.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, 2, 1, padding)
    need = lib().stem_conv_bwd_weight_workspace(N, Ho, Wo, Cout)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((28, Cout), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    gp, ldg = _pl(g)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    lib().stem_conv_bwd_weight_slabs_bn(xp, ldx, gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift), act, _p(bn.mean), _p(bn.invstd),
                                        _p(bn.coef), _p(ws), need, ctypes.byref(rows), N, H, W, Cout, pt, pl, Ho, Wo, _stream())
    lib().reduce_rows(_p(ws), rows.value, 28 * Cout, _p(gw), 0, _stream())
    return gw[:27].reshape(3, 3, 3, Cout)


def col2im(gcol, x_shape, k, stride=1, rate=1, padding='same', out=None, accumulate=False):
    N, H, W, Cin = x_shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=gcol.device)
    gp, ldg = _pl(gx)
    lib().col2im(_p(gcol), gcol.shape[-1], gp, ldg, int(accumulate), N, H, W, Cin, k, stride, rate, pt, pl, Ho, Wo,
                 _stream())
    return gx


def scale_bcast_fwd(x, s, scale=None, shift=None, act=ACT_NONE, s_act=ACT_NONE, out=None):
    """y = act(x*scale+shift) * s_act(s)[n]  with s (N,1,1,C): the SE-block Multiply"""
    N, H, W, C = x.shape
    y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    sp, lds = _pl(s)
    yp, ldy = _pl(y)
    lib().scale_bcast_fwd(xp, ldx, _p(scale), _p(shift), act, sp, lds, s_act, yp, ldy, N, H * W, C, _stream())
    return y


def scale_bcast_bwd(gy, x, s, scale=None, shift=None, act=ACT_NONE, s_act=ACT_NONE, chunked=True):
    N, H, W, C = x.shape
    gx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    gs = torch.empty((N, 1, 1, C), dtype=torch.float32, device=x.device)
    gp, ldg = _pl(gy)
    xp, ldx = _pl(x)
    sp, lds = _pl(s)
    ws, wsb = _pool_ws(N, H * W, C, x.device, chunked)
    lib().scale_bcast_bwd(gp, ldg, xp, ldx, _p(scale), _p(shift), act, sp, lds, s_act, _p(gx), C, 0, _p(gs), C, N, H * W, C,
                          _p(ws), wsb, _stream())
    return gx, gs


# ------------------------------------------------------------------------------------- mixed precision (bf16 storage)
# wrappers over the *_bf16 entry points: activation tensors are torch.bfloat16 (or float32 where the C ABI takes an
# `*_is_f32` flag), BatchNorm coefficients / partial rows / weight gradients float32
def _plb(t):
    """(device pointer, row stride in ELEMENTS, is_f32) of a bf16 (or f32) activation tensor or channel-slice view"""
    if t is None:
        return None, 0, 0
    if not t.is_cuda or t.dtype not in (torch.bfloat16, torch.float32):
        raise Dl3pError('bf16 ops need bfloat16 / float32 tensors on the HIP device (got %s on %s)' % (t.dtype, t.device))
    if t.stride(-1) != 1:
        raise Dl3pError('channel dim must be contiguous')
    ld = t.stride(-2) if t.dim() >= 2 else t.shape[-1]
    return t.data_ptr(), ld, int(t.dtype == torch.float32)


def _bf(t):
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


def pwconv_fwd_bf16(x, w, bias=None, in_scale=None, in_shift=None, in_act=ACT_NONE, partials=None, out_f32=False):
    """x (.., K) bf16 / f32, w (K, N) float32 master kernel -> y (.., N) bf16 (f32 with out_f32); (y, rows) with partials"""
    K, Nn = w.shape
    M = _rows(x)
    wt = _bf(w.t().contiguous())
    y = torch.empty(tuple(x.shape[:-1]) + (Nn,), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    xp, ldx, xf = _plb(x)
    yp, ldy, yf = _plb(y)
    rows = ctypes.c_int(0)
    lib().pwconv_fwd_bf16(xp, ldx, xf, _p(in_scale), _p(in_shift), in_act, _p(wt), _p(bias), yp, ldy, yf, _p(partials),
                          ctypes.byref(rows), M, K, Nn, _stream())
    return (y, rows.value) if partials is not None else y


def pwconv_bwd_data_bf16(dy, w, out=None, accumulate=False):
    K, Nn = w.shape
    M = _rows(dy)
    gx = out if out is not None else torch.empty(tuple(dy.shape[:-1]) + (K,), dtype=torch.bfloat16, device=dy.device)
    dp, ldd, df = _plb(dy)
    gp, ldg, _ = _plb(gx)
    lib().pwconv_bwd_data_bf16(dp, ldd, df, _p(_bf(w.contiguous())), gp, ldg, int(accumulate), M, K, Nn, _stream())
    return gx


def pwconv_bwd_data_bn_bf16(dy, w, z, scale, shift, act, mean, invstd, partials, out=None, accumulate=False):
    """bf16 data gradient + the BatchNorm-backward partial sums of the BatchNorm behind it -> (gx, rows)"""
    K, Nn = w.shape
    M = _rows(dy)
    gx = out if out is not None else torch.empty(tuple(dy.shape[:-1]) + (K,), dtype=torch.bfloat16, device=dy.device)
    dp, ldd, _ = _plb(dy)
    gp, ldg, _ = _plb(gx)
    zp, ldz, _ = _plb(z)
    rows = ctypes.c_int(0)
    lib().pwconv_bwd_data_bn_bf16(dp, ldd, _p(_bf(w.contiguous())), gp, ldg, int(accumulate), M, K, Nn, zp, ldz, _p(scale),
                                  _p(shift), act, _p(mean), _p(invstd), _p(partials), ctypes.byref(rows), _stream())
    return gx, rows.value


def pwconv_bwd_weight_bf16(x, dy, in_scale=None, in_shift=None, in_act=ACT_NONE, with_bias=False):
    K, Nn = x.shape[-1], dy.shape[-1]
    M = _rows(x)
    need = lib().pwconv_bwd_weight_workspace_bf16(M, K, Nn)
    ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((K, Nn), dtype=torch.float32, device=x.device)
    gb = torch.empty((Nn,), dtype=torch.float32, device=x.device) if with_bias else None
    xp, ldx, _ = _plb(x)
    dp, ldd, df = _plb(dy)
    lib().pwconv_bwd_weight_bf16(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, df, _p(gw), _p(gb), _p(ws),
                                 ws.numel() * 4, M, K, Nn, _stream())
    return (gw, gb) if with_bias else gw


def dwconv2d_fwd_bf16(x, w, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE, partials=None):
    N, H, W, C = x.shape
    k = w.shape[0]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    y = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    xp, ldx, _ = _plb(x)
    yp, ldy, _ = _plb(y)
    rows = ctypes.c_int(0)
    lib().dwconv2d_fwd_bf16(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(_bf(w.contiguous())), yp, ldy, _p(partials),
                            ctypes.byref(rows), N, H, W, C, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def dwconv2d_bwd_data_bf16(dy, w, x_shape, stride=1, rate=1, padding='same', out=None, accumulate=False):
    N, H, W, C = x_shape
    k = w.shape[0]
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty((N, H, W, C), dtype=torch.bfloat16, device=dy.device)
    dp, ldd, _ = _plb(dy)
    gp, ldg, _ = _plb(gx)
    lib().dwconv2d_bwd_data_bf16(dp, ldd, _p(_bf(w.contiguous())), gp, ldg, int(accumulate), N, H, W, C, k, stride, rate,
                                 pt, pl, Ho, Wo, _stream())
    return gx


def dwconv2d_bwd_weight_bf16(x, dy, k, stride=1, rate=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE):
    N, H, W, C = x.shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    need = lib().dwconv2d_bwd_weight_workspace_bf16(N, Ho, Wo, C, k)
    ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((k, k, C), dtype=torch.float32, device=x.device)
    xp, ldx, _ = _plb(x)
    dp, ldd, _ = _plb(dy)
    lib().dwconv2d_bwd_weight_bf16(xp, ldx, _p(in_scale), _p(in_shift), in_act, dp, ldd, _p(gw), _p(ws), ws.numel() * 4,
                                   N, H, W, C, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return gw


def bn_backward_bf16(bn, g, z, act, partials):
    """in-place BatchNorm backward on bf16 tensors (reduce + finalize + apply); bn: BNState with scale/shift/mean/invstd"""
    M = _rows(z)
    gp, ldg, _ = _plb(g)
    zp, ldz, _ = _plb(z)
    rows = ctypes.c_int(0)
    lib().bn_bwd_reduce_bf16(gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift), act, _p(bn.mean), _p(bn.invstd), _p(partials),
                             ctypes.byref(rows), M, bn.C, _stream())
    lib().bn_bwd_finalize(_p(partials), rows.value, None, bn.C, float(M), _p(bn.gamma), _p(bn.invstd), _p(bn.scale), 0,
                          _p(bn.dgamma), _p(bn.dbeta), _p(bn.coef), _stream())
    lib().bn_bwd_apply_bf16(gp, ldg, zp, ldz, _p(bn.scale), _p(bn.shift), act, _p(bn.mean), _p(bn.invstd), _p(bn.coef),
                            gp, ldg, 0, M, bn.C, _stream())
    return g


def affine_act_bf16(x, scale=None, shift=None, act=ACT_NONE, residual=None, rscale=None, rshift=None, ract=ACT_NONE):
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    xp, ldx, _ = _plb(x)
    rp, ldr, _ = _plb(residual)
    yp, ldy, _ = _plb(y)
    lib().affine_act_bf16(xp, ldx, _p(scale), _p(shift), act, rp, ldr, _p(rscale), _p(rshift), ract, 0.0, 0, None, yp, ldy,
                          _rows(x), x.shape[-1], _stream())
    return y


def global_avgpool_fwd_bf16(x, in_scale=None, in_shift=None, in_act=ACT_NONE, out_scale=1.0):
    N, H, W, C = x.shape
    y = torch.empty((N, 1, 1, C), dtype=torch.bfloat16, device=x.device)
    ws = torch.zeros(lib().pool_workspace_bf16(N, H * W, C) // 4 + 4, dtype=torch.float32, device=x.device)
    xp, ldx, _ = _plb(x)
    lib().global_avgpool_fwd_bf16(xp, ldx, _p(in_scale), _p(in_shift), in_act, y.data_ptr(), C, float(out_scale), N, H * W, C,
                                  _p(ws), ws.numel() * 4, _stream())
    return y


def scale_bcast_fwd_bf16(x, s, scale=None, shift=None, act=ACT_NONE, s_act=ACT_NONE):
    N, H, W, C = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    xp, ldx, _ = _plb(x)
    lib().scale_bcast_fwd_bf16(xp, ldx, _p(scale), _p(shift), act, s.data_ptr(), C, s_act, y.data_ptr(), C, N, H * W, C, _stream())
    return y


def scale_bcast_bwd_bf16(gy, x, s, scale=None, shift=None, act=ACT_NONE, s_act=ACT_NONE):
    N, H, W, C = x.shape
    gx = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    gs = torch.empty((N, 1, 1, C), dtype=torch.bfloat16, device=x.device)
    ws = torch.zeros(lib().pool_workspace_bf16(N, H * W, C) // 4 + 4, dtype=torch.float32, device=x.device)
    gp, ldg, _ = _plb(gy)
    xp, ldx, _ = _plb(x)
    lib().scale_bcast_bwd_bf16(gp, ldg, xp, ldx, _p(scale), _p(shift), act, s.data_ptr(), C, s_act, gx.data_ptr(), C, 0,
                               gs.data_ptr(), C, N, H * W, C, _p(ws), ws.numel() * 4, _stream())
    return gx, gs


def resize_bilinear_fwd_bf16(x, H, W):
    N, h, w, C = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    xp, ldx, _ = _plb(x)
    lib().resize_bilinear_fwd_bf16(xp, ldx, y.data_ptr(), C, N, h, w, C, H, W, _stream())
    return y


def resize_bilinear_bwd_bf16(gy, h, w):
    N, H, W, C = gy.shape
    gx = torch.empty((N, h, w, C), dtype=torch.bfloat16, device=gy.device)
    gp, ldg, _ = _plb(gy)
    lib().resize_bilinear_bwd_bf16(gp, ldg, gx.data_ptr(), C, 0, N, h, w, C, H, W, _stream())
    return gx


# ------------------------------------------------------------------------------------- split-bf16 GEMMs (fp32-accurate)
def split_bf16x3(mat):
    """mat (rows, cols) float32 -> (3, rows, pitch) int16: the exact 3-way bf16 split of every element (planes hi, mid, lo),
    pitch = cols rounded up to 32, zero padded -- the pre-split B operand of the *_sb entry points"""
    rows, cols = mat.shape
    assert mat.is_contiguous() and mat.dtype == torch.float32
    pitch = (cols + 31) // 32 * 32
    out = torch.empty((3, rows, pitch), dtype=torch.int16, device=mat.device)
    table = torch.tensor([[0, rows, cols, cols, 0, pitch]], dtype=torch.int64, device=mat.device)
    lib().split_bf16x3_batch(_p(mat), _p(out), _p(table), 1, _stream())
    return out


def pwconv_fwd_sb(x, wt_sp, K, bias=None, in_scale=None, in_shift=None, in_act=ACT_NONE, out=None, partials=None):
    """pwconv_fwd_wt on the bf16 matrix pipe; wt_sp = split_bf16x3(wt) with wt (N, K)"""
    M = _rows(x)
    N, pitch = wt_sp.shape[1], wt_sp.shape[2]
    y = out if out is not None else torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().pwconv_fwd_sb(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(wt_sp), pitch, _p(bias), yp, ldy, _p(partials),
                        ctypes.byref(rows), M, K, N, _stream())
    return (y, rows.value) if partials is not None else y


def pwconv_bwd_data_sb(dy, w_sp, N, out=None, accumulate=False, z=None, scale=None, shift=None, act=ACT_NONE, mean=None,
                       invstd=None, partials=None):
    """pwconv_bwd_data (z is None) / pwconv_bwd_data_bn on the bf16 matrix pipe; w_sp = split_bf16x3(w) with w (K, N)"""
    M = _rows(dy)
    K, pitch = w_sp.shape[1], w_sp.shape[2]
    gx = out if out is not None else torch.empty(tuple(dy.shape[:-1]) + (K,), dtype=torch.float32, device=dy.device)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    lib().pwconv_bwd_data_sb(dp, ldd, _p(w_sp), pitch, gp, ldg, int(accumulate), M, K, N, zp, ldz or 0, _p(scale), _p(shift), act,
                             _p(mean), _p(invstd), _p(partials), ctypes.byref(rows), _stream())
    return (gx, rows.value) if z is not None else gx


def pwconv_bwd_data_sb_apply(g, z_out, bn_scale, bn_shift, bn_act, bn_mean, bn_invstd, bn_coef, w_sp, N, dz=None, out=None,
                             accumulate=False, z=None, scale=None, shift=None, act=ACT_NONE, mean=None, invstd=None, partials=None):
    """pwconv_bwd_data_sb with the BatchNorm-backward apply of (g, z_out) folded into its staged operand; -> (dz, gx[, rows]):
    dz = bn_coef[0] * (g * act'(z_out * bn_scale + bn_shift) - bn_coef[1] - xhat * bn_coef[2]) is written to `dz` (default: over g)"""
    M = _rows(g)
    K, pitch = w_sp.shape[1], w_sp.shape[2]
    gx = out if out is not None else torch.empty(tuple(g.shape[:-1]) + (K,), dtype=torch.float32, device=g.device)
    dzt = dz if dz is not None else g
    gp_, ldg_ = _pl(g)
    zo, ldzo = _pl(z_out)
    dzp, lddz = _pl(dzt)
    gxp, ldgx = _pl(gx)
    zp, ldz = _pl(z)
    rows = ctypes.c_int(0)
    lib().pwconv_bwd_data_sb_apply(gp_, ldg_, zo, ldzo, _p(bn_scale), _p(bn_shift), bn_act, _p(bn_mean), _p(bn_invstd), _p(bn_coef),
                                   dzp, lddz, _p(w_sp), pitch, gxp, ldgx, int(accumulate), M, K, N, zp, ldz or 0, _p(scale), _p(shift),
                                   act, _p(mean), _p(invstd), _p(partials), ctypes.byref(rows), _stream())
    return (dzt, gx, rows.value) if z is not None else (dzt, gx)


def col2im_bf16(gcol, x_shape, k, stride=1, rate=1, padding='same', out=None, accumulate=False):
    """bf16 twin of col2im: gcol (N,Ho,Wo,kp) bf16 -> gx (N,H,W,Cin) bf16"""
    N, H, W, Cin = x_shape
    Ho, Wo, pt, pl = conv_geometry(H, W, k, stride, rate, padding)
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.bfloat16, device=gcol.device)
    gp, ldg, _ = _plb(gx)
    lib().col2im_bf16(_p(gcol), gcol.shape[-1], gp, ldg, int(accumulate), N, H, W, Cin, k, stride, rate, pt, pl, Ho, Wo, _stream())
    return gx


def maxpool2d_fwd_bf16(x, k, stride, pad, in_scale=None, in_shift=None, in_act=ACT_NONE, argmax=None):
    N, H, W, C = x.shape
    pt, pb, pl, pr = pad
    Ho, Wo = (H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1
    y = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    xp, ldx, _ = _plb(x)
    yp, ldy, _ = _plb(y)
    lib().maxpool2d_fwd_bf16(xp, ldx, _p(in_scale), _p(in_shift), in_act, yp, ldy, _p(argmax), N, H, W, C, k, stride, pt, pl, Ho,
                             Wo, _stream())
    return y


def maxpool2d_bwd_bf16(dy, argmax, x_shape, k, stride, pad, out=None, accumulate=False):
    N, H, W, C = x_shape
    pt, pb, pl, pr = pad
    Ho, Wo = dy.shape[1], dy.shape[2]
    gx = out if out is not None else torch.empty(x_shape, dtype=torch.bfloat16, device=dy.device)
    dp, ldd, _ = _plb(dy)
    gp, ldg, _ = _plb(gx)
    lib().maxpool2d_bwd_bf16(dp, ldd, _p(argmax), gp, ldg, int(accumulate), N, H, W, C, k, stride, pt, pl, Ho, Wo, _stream())
    return gx


# ------------------------------------------------------------------------------------- fused inverted-residual block
def irb_supported(x_shape, C, stride=1, padding='same', backward=False):
    N, H, W, K = x_shape
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, stride, 1, padding)
    fn = lib().irb_bwd_supported if backward else lib().irb_supported
    return bool(fn(N, H, W, K, C, 3, stride, 1, pt, pl, Ho, Wo))


def irb_cov_sums(x, in_scale=None, in_shift=None, in_act=ACT_NONE):
    """float64 [K + K*K]: sum over pixels of the (prologue-applied) input and of its outer product"""
    K = x.shape[-1]
    M = _rows(x)
    xp, ldx = _pl(x)
    cap = lib().irb_cov_rows_max()
    rows_buf = torch.empty(cap * (K + K * K), dtype=torch.float64, device=x.device)
    rows = ctypes.c_int(0)
    lib().irb_cov_stats(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(rows_buf), cap, ctypes.byref(rows), M, K, _stream())
    sums = torch.empty(K + K * K, dtype=torch.float64, device=x.device)
    lib().irb_cov_reduce(_p(rows_buf), rows.value, K, _p(sums), _stream())
    return sums


def irb_bn_finalize_cov(bn, sums, w1, count, update_moving=True):
    """the expand BatchNorm's coefficients from the covariance sums of the expand conv's INPUT; w1 (K, C)"""
    K, C = w1.shape[-2], w1.shape[-1]
    lib().irb_bn_finalize_cov(_p(sums), _p(w1), K, C, float(count), _p(bn.gamma), _p(bn.beta), bn.eps, bn.momentum,
                              _p(bn.moving_mean), _p(bn.moving_var), int(update_moving), _p(bn.scale), _p(bn.shift),
                              _p(bn.mean), _p(bn.invstd), _stream())


def irb_fwd(x, w1, bn_scale, bn_shift, bn_act, wdw, stride=1, padding='same', in_scale=None, in_shift=None,
            in_act=ACT_NONE, out=None, partials=None):
    """x (N,H,W,K); w1 (K,C); wdw (3,3,C) -> raw depthwise output (N,Ho,Wo,C) [, rows]"""
    N, H, W, K = x.shape
    C = w1.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, stride, 1, padding)
    y = out if out is not None else torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    yp, ldy = _pl(y)
    rows = ctypes.c_int(0)
    lib().irb_fwd(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w1), _p(bn_scale), _p(bn_shift), bn_act, _p(wdw), yp, ldy,
                  _p(partials), ctypes.byref(rows), N, H, W, K, C, stride, pt, pl, Ho, Wo, _stream())
    return (y, rows.value) if partials is not None else y


def irb_bwd_sums(x, w1, bn, bn_act, wdw, dy, stride=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE):
    """pass A -> (depthwise kernel gradient (3,3,C), BatchNorm-backward sums float64 [2][C])"""
    N, H, W, K = x.shape
    C = w1.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, stride, 1, padding)
    nbytes = lib().irb_bwd_workspace(0, N, H, W, K, C, stride, pt, pl)
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    part = torch.empty(MAX_STAT_ROWS * 2 * C, dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    rows = ctypes.c_int(0)
    lib().irb_bwd_sums(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w1), _p(bn.scale), _p(bn.shift), bn_act, _p(bn.mean),
                       _p(bn.invstd), _p(wdw), dp, ldd, _p(slabs), nbytes, ctypes.byref(rows), _p(part), N, H, W, K, C, stride,
                       pt, pl, Ho, Wo, _stream())
    r = rows.value
    gw = slabs[:r * 9 * C].view(r, 9 * C).double().sum(0).float().view(3, 3, C)
    return gw, part, r


def irb_bwd_data(x, w1, bn, bn_act, wdw, dy, stride=1, padding='same', in_scale=None, in_shift=None, in_act=ACT_NONE,
                 out=None, accumulate=False, want_gx=True, front=None):
    """pass B -> (expand kernel gradient (K,C), gx, [front BatchNorm partial rows, rows]); bn.coef must hold the triple;
    front = (z0, scale0, shift0, act0, mean0, invstd0)"""
    N, H, W, K = x.shape
    C = w1.shape[-1]
    Ho, Wo, pt, pl = conv_geometry(H, W, 3, stride, 1, padding)
    nbytes = lib().irb_bwd_workspace(1, N, H, W, K, C, stride, pt, pl)
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    gx = None
    if want_gx:
        gx = out if out is not None else torch.empty((N, H, W, K), dtype=torch.float32, device=x.device)
    xp, ldx = _pl(x)
    dp, ldd = _pl(dy)
    gp, ldg = _pl(gx)
    rows = ctypes.c_int(0)
    part0 = None
    fz = (None, 0, None, None, ACT_NONE, None, None)
    if front is not None:
        z0, s0, h0, a0, m0, i0 = front
        zp, ldz = _pl(z0)
        fz = (zp, ldz, _p(s0), _p(h0), a0, _p(m0), _p(i0))
        part0 = torch.empty(MAX_STAT_ROWS * 2 * K, dtype=torch.float32, device=x.device)
    lib().irb_bwd_data(xp, ldx, _p(in_scale), _p(in_shift), in_act, _p(w1), _p(bn.scale), _p(bn.shift), bn_act, _p(bn.mean),
                       _p(bn.invstd), _p(bn.coef), _p(wdw), dp, ldd, _p(slabs), nbytes, ctypes.byref(rows), gp, ldg,
                       int(accumulate), *fz, _p(part0), N, H, W, K, C, stride, pt, pl, Ho, Wo, _stream())
    r = rows.value
    gw = slabs[:r * K * C].view(r, K * C).double().sum(0).float().view(K, C)
    return gw, gx, part0, r
