"""Mixed-precision path (BASELINE.json configs[4]: MobileNetV3-Large, bf16; reference switch train.py:37-46).

Oracle leg: the fp64 oracle with `net.bf16 = True` rounds to bfloat16 at the storage points of the HIP path (oracle/np_net.py
Net.bf16, csrc/bf16.h).  Tolerances (DESIGN.md "bf16"): an op reproduces the rounded fp64 result to ONE bf16 ulp per
stored element (2^-8 relative: fp32 vs fp64 accumulation can land on the other side of a rounding boundary) -- checked as
|got - want| <= 2^-7 |want| + eps; fp32 outputs (weight gradients, statistics) to 2e-3 of the tensor's scale; through a
whole model the logits to 3e-2 of their range and every parameter gradient to a cosine similarity > 0.99 with the oracle
whose BACKWARD runs unrounded."""
import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import np_ops as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
Q = O.bf16_round


NOISE_ULPS = 16.0
ACT_FRAC, ACT_REL = 0.995, 6e-3      # activation gradients: fraction of elements within two ulps, relative L2 (see the test)


def _record_bf16_backward(rec):
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'bf16_backward_parity.jsonl'), 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass


def TB(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16)


def TF(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def np64(t):
    if not torch.is_tensor(t):
        return np.asarray(t, np.float64)
    return t.detach().float().cpu().numpy().astype(np.float64)


def close_bf16(got, want, what, slack=1.0):
    got, want = np64(got), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    # one bf16 ulp of the element + the echo of one-ulp flips among the operands (a sum of K terms moves by ~1e-3 of its scale)
    tol = slack * (2.0 ** -7) * np.abs(want) + slack * 1e-3 * float(np.abs(want).max()) + 1e-30
    bad = np.abs(got - want) > tol
    assert not bad.any(), '%s: %d of %d outside one bf16 ulp, worst %g' % (what, bad.sum(), bad.size, np.abs(got - want).max())


def close_f32(got, want, what, rtol=2e-3):
    got, want = np64(got), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.abs(got - want).max() <= rtol * max(1e-6, float(np.abs(want).max())), (what, np.abs(got - want).max(), np.abs(want).max())


PW = [(2 * 33 * 33, 320, 256), (1000, 24, 72), (777, 72, 24), (4101, 16, 64), (513, 304, 256), (300, 960, 160), (131, 184, 80),
      (64 * 128 + 5, 160, 960), (3, 240, 64), (1, 960, 240), (257, 1280, 256), (5000, 256, 24),
      # BASELINE configs[4] launch shapes (MobileNetV3-Large 1024 x 2048, batch 1): the decoder GEMMs on the wave-streaming
      # kernel, the 512 x 1024 stem-side layers, the 64 x 128 body and ASPP
      (256 * 512, 304, 256), (256 * 512, 256, 256), (512 * 1024, 16, 64), (128 * 256, 120, 40), (64 * 128, 960, 160),
      (64 * 128, 1280, 256), (256 * 512, 256, 24)]


@pytest.mark.parametrize('case', PW)
def test_pointwise_bf16(ops, case):
    M, K, Nn = case
    rng = np.random.default_rng(M + 7 * K + Nn)
    x = Q(rng.standard_normal((M, K)))
    w = rng.standard_normal((K, Nn)) / np.sqrt(K)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = (rng.standard_normal(K) * 0.3).astype(np.float32)
    b = (rng.standard_normal(Nn) * 0.2).astype(np.float32)
    wq = Q(w)
    a = Q(O.act_fwd(np.float32(np.float64(x) * sc + sh), O.ACT_HSWISH).astype(np.float64))      # fmaf, then fp32 activation
    y_ref = a @ wq
    part = ops.new_partials(Nn, DEV)
    y, rows = ops.pwconv_fwd_bf16(TB(x), TF(w), None, TF(sc), TF(sh), ops.ACT_HSWISH, partials=part)
    close_bf16(y, Q(y_ref), 'forward')
    yq = np64(y)
    p = part[:rows * 2 * Nn].reshape(rows, 2, Nn).double().sum(0).cpu().numpy()
    close_f32(p[0], yq.sum(0), 'stat sum (of the stored values)', 1e-4 + 1e-5 * np.sqrt(M))
    close_f32(p[1], (yq ** 2).sum(0), 'stat sum of squares', 1e-4)
    # bias, no prologue, fp32 output (the logits layer)
    y32 = ops.pwconv_fwd_bf16(TB(x), TF(w), TF(b), out_f32=True)
    close_f32(y32, x @ wq + b, 'forward fp32 out', 1e-5)
    gy = Q(rng.standard_normal((M, Nn)))
    gx = ops.pwconv_bwd_data_bf16(TB(gy), TF(w))
    close_bf16(gx, Q(gy @ wq.T), 'data gradient')
    base = Q(rng.standard_normal((M, K)))
    gx2 = ops.pwconv_bwd_data_bf16(TB(gy), TF(w), out=TB(base), accumulate=True)
    close_bf16(gx2, Q(Q(gy @ wq.T).astype(np.float64) * 0 + gy @ wq.T + base), 'data gradient accumulate', slack=2.0)
    gx3 = ops.pwconv_bwd_data_bf16(TF(gy), TF(w))            # fp32 gradient in (the logits layer): rounded on the way in
    close_bf16(gx3, Q(gy @ wq.T), 'data gradient from fp32')
    gw, gb = ops.pwconv_bwd_weight_bf16(TB(x), TB(gy), TF(sc), TF(sh), ops.ACT_HSWISH, with_bias=True)
    close_f32(gw, a.T @ gy, 'weight gradient', 2e-5 * np.sqrt(M) + 1e-5)
    close_f32(gb, gy.sum(0), 'bias gradient', 1e-4)


@pytest.mark.parametrize('kg', [1, 2, 4])
@pytest.mark.parametrize('case', [(64 * 128, 960, 160), (2000, 672, 112), (1003, 1288, 264), (513, 200, 80), (8192, 80, 480)])
def test_pointwise_bf16_k_groups(ops, case, kg):
    """pwb_gemm with its K groups pinned (dl3p_set_option("bf16_kg")): one group (the plain tiled kernel), two and four groups of
    four waves taking every second / fourth K-step of a tile and summing their accumulators through LDS -- a reduction that is not
    a multiple of 32 * groups, row and column tails, forward with prologue + statistics and both data gradients"""
    M, K, Nn = case
    L = ops.lib()
    L.set_option(b'bf16_kg', kg)
    try:
        rng = np.random.default_rng(M + 7 * K + Nn)
        x = Q(rng.standard_normal((M, K)))
        w = rng.standard_normal((K, Nn)) / np.sqrt(K)
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
        sh = (rng.standard_normal(K) * 0.3).astype(np.float32)
        wq = Q(w)
        a = Q(O.act_fwd(np.float32(np.float64(x) * sc + sh), O.ACT_RELU6).astype(np.float64))
        part = ops.new_partials(Nn, DEV)
        y, rows = ops.pwconv_fwd_bf16(TB(x), TF(w), None, TF(sc), TF(sh), ops.ACT_RELU6, partials=part)
        close_bf16(y, Q(a @ wq), 'forward')
        yq = np64(y)
        p = part[:rows * 2 * Nn].reshape(rows, 2, Nn).double().sum(0).cpu().numpy()
        close_f32(p[0], yq.sum(0), 'stat sum (of the stored values)', 1e-4 + 1e-5 * np.sqrt(M))
        close_f32(p[1], (yq ** 2).sum(0), 'stat sum of squares', 1e-4)
        close_bf16(ops.pwconv_fwd_bf16(TB(x), TF(w)), Q(np.float64(x) @ wq), 'forward, no prologue, no statistics')
        gy = Q(rng.standard_normal((M, Nn)))
        gx = ops.pwconv_bwd_data_bf16(TB(gy), TF(w))
        close_bf16(gx, Q(gy @ wq.T), 'data gradient')
        z = Q(rng.standard_normal((M, K)) * 1.5 + 0.4)
        mu = (rng.standard_normal(K) * 0.2).astype(np.float32)
        inv = rng.uniform(0.5, 2.0, K).astype(np.float32)
        part2 = ops.new_partials(K, DEV)
        gx2, rows2 = ops.pwconv_bwd_data_bn_bf16(TB(gy), TF(w), TB(z), TF(sc), TF(sh), O.ACT_RELU6, TF(mu), TF(inv), part2)
        assert torch.equal(gx2, gx), 'the sums epilogue changes the gradient'
        u = np.float32(np.float64(z) * sc + sh)
        g = np64(gx2) * O.act_bwd(u.astype(np.float64), np.ones((M, K)), O.ACT_RELU6)
        p2 = part2[:rows2 * 2 * K].reshape(rows2, 2, K).double().sum(0).cpu().numpy()
        close_f32(p2[0], g.sum(0), 'backward sum', 2e-5 * np.sqrt(M) + 1e-5)
        close_f32(p2[1], (g * ((np.float64(z) - mu) * inv)).sum(0), 'backward sum with xhat', 2e-5 * np.sqrt(M) + 1e-5)
    finally:
        L.set_option(b'bf16_kg', -1)


@pytest.mark.parametrize('case', [(2 * 33 * 33, 320, 256), (1000, 24, 72), (777, 72, 24), (4101, 16, 64), (513, 304, 256),
                                  (300, 960, 160), (64 * 128 * 9 + 5, 160, 304), (70003, 64, 24), (66000, 256, 144)])
@pytest.mark.parametrize('act', [O.ACT_RELU6, O.ACT_NONE, O.ACT_HSWISH])
def test_pointwise_bf16_data_gradient_with_bn_sums(ops, case, act):
    """dl3p_pwconv_bwd_data_bn_bf16: the data gradient of dl3p_pwconv_bwd_data_bf16 bit for bit, and the partial sums
    dl3p_bn_bwd_reduce_bf16 forms from the stored gradient (tiled and streaming kernels, accumulate)"""
    M, K, Nn = case
    if act != O.ACT_RELU6 and M > 5000:
        pytest.skip('large shapes once')
    rng = np.random.default_rng(M + K + Nn)
    gy = Q(rng.standard_normal((M, Nn)))
    w = rng.standard_normal((K, Nn)) / np.sqrt(Nn)
    z = Q(rng.standard_normal((M, K)) * 1.5 + 0.4)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = (rng.standard_normal(K) * 0.3).astype(np.float32)
    mu = (rng.standard_normal(K) * 0.2).astype(np.float32)
    inv = rng.uniform(0.5, 2.0, K).astype(np.float32)
    base = Q(rng.standard_normal((M, K)))
    for accumulate in (False, True):
        part = ops.new_partials(K, DEV)
        out0 = TB(base) if accumulate else None
        out1 = TB(base) if accumulate else None
        gx_ref = ops.pwconv_bwd_data_bf16(TB(gy), TF(w), out=out0, accumulate=accumulate)
        gx, rows = ops.pwconv_bwd_data_bn_bf16(TB(gy), TF(w), TB(z), TF(sc), TF(sh), act, TF(mu), TF(inv), part, out=out1,
                                               accumulate=accumulate)
        assert torch.equal(gx, gx_ref), 'the sums epilogue changes the gradient'
        g = np64(gx)
        u = np.float32(np.float64(z) * sc + sh)
        d = g * O.act_bwd(u.astype(np.float64), np.ones_like(g), act)
        xh = (np.float64(z) - mu) * inv
        p = part[:rows * 2 * K].reshape(rows, 2, K).double().sum(0).cpu().numpy()
        close_f32(p[0], d.sum(0), 'sum g\'', 2e-5 * np.sqrt(M) + 1e-5)
        close_f32(p[1], (d * xh).sum(0), 'sum g\' xhat', 2e-5 * np.sqrt(M) + 1e-5)


DW = [(2, 33, 33, 160, 3, 1, 18), (2, 33, 33, 160, 3, 1, 6), (1, 65, 47, 72, 3, 2, 1), (2, 16, 20, 24, 3, 2, 1), (1, 16, 24, 40, 5, 1, 2),
      (1, 16, 24, 72, 5, 2, 1), (2, 33, 33, 960, 5, 1, 2), (1, 7, 6, 24, 5, 1, 1), (1, 64, 96, 16, 3, 1, 1), (3, 9, 9, 12, 3, 1, 1),
      # configs[4] launch shapes: decoder 256 x 512 x 304, 5 x 5 rate 2 on 64 x 128 x 960, the three ASPP rates, a stride-2 layer
      (1, 256, 512, 304, 3, 1, 1), (1, 64, 128, 960, 5, 1, 2), (1, 64, 128, 160, 3, 1, 6), (1, 64, 128, 160, 3, 1, 12),
      (1, 64, 128, 160, 3, 1, 18), (1, 256, 512, 72, 3, 2, 1), (1, 128, 256, 120, 5, 1, 1)]


@pytest.mark.parametrize('case', DW)
def test_depthwise_bf16(ops, case):
    N, H, W, C, k, s, r = case
    rng = np.random.default_rng(H * 5 + C + k + r)
    x = Q(rng.standard_normal((N, H, W, C)))
    w = rng.standard_normal((k, k, C)) * 0.3
    sc = rng.uniform(0.5, 1.5, C).astype(np.float32)
    sh = (rng.standard_normal(C) * 0.3).astype(np.float32)
    wq = Q(w)
    a = Q(O.act_fwd(np.float32(np.float64(x) * sc + sh), O.ACT_RELU6).astype(np.float64))
    y_ref = O.dwconv2d_fwd(a, wq, s, r, 'same')
    part = ops.new_partials(C, DEV)
    y, rows = ops.dwconv2d_fwd_bf16(TB(x), TF(w), s, r, 'same', TF(sc), TF(sh), ops.ACT_RELU6, partials=part)
    close_bf16(y, Q(y_ref), 'forward')
    yq = np64(y).reshape(-1, C)
    p = part[:rows * 2 * C].reshape(rows, 2, C).double().sum(0).cpu().numpy()
    close_f32(p[0], yq.sum(0), 'stat sum', 1e-4 + 1e-5 * np.sqrt(yq.shape[0]))
    close_f32(p[1], (yq ** 2).sum(0), 'stat sum of squares', 1e-4)
    gy = Q(rng.standard_normal(y_ref.shape))
    gx_ref, gw_ref = O.dwconv2d_bwd(a, wq, gy, s, r, 'same')
    gx = ops.dwconv2d_bwd_data_bf16(TB(gy), TF(w), (N, H, W, C), s, r, 'same')
    close_bf16(gx, Q(gx_ref), 'data gradient')
    gw = ops.dwconv2d_bwd_weight_bf16(TB(x), TB(gy), k, s, r, 'same', TF(sc), TF(sh), ops.ACT_RELU6)
    close_f32(gw, gw_ref, 'weight gradient', 1e-4)


def test_elementwise_bf16(ops):
    rng = np.random.default_rng(4)
    N, H, W, C = 2, 17, 19, 72
    z = Q(rng.standard_normal((N, H, W, C)) * 2 + 0.3)
    g = Q(rng.standard_normal((N, H, W, C)))
    gamma, beta = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.2
    y_ref, cache, _ = O.bn_train_fwd(z, gamma, beta, 1e-3)
    bn = ops.BNState(C, DEV, 1e-3)
    bn.gamma.copy_(TF(gamma)); bn.beta.copy_(TF(beta))
    mean, var = z.reshape(-1, C).mean(0), z.reshape(-1, C).var(0)
    invstd = 1 / np.sqrt(var + 1e-3)
    bn.mean.copy_(TF(mean)); bn.invstd.copy_(TF(invstd))
    bn.scale.copy_(TF(gamma * invstd)); bn.shift.copy_(TF(beta - mean * gamma * invstd))
    dz_ref, gg, gb = O.bn_train_bwd(O.act_bwd(y_ref, g, O.ACT_RELU6), cache)
    gt = TB(g)
    ops.bn_backward_bf16(bn, gt, TB(z), ops.ACT_RELU6, ops.new_partials(C, DEV))
    close_bf16(gt, Q(dz_ref), 'bn backward dz', slack=2.0)
    close_f32(bn.dgamma, gg, 'dgamma', 1e-4)
    close_f32(bn.dbeta, gb, 'dbeta', 1e-4)
    # materialise + residual
    r = Q(rng.standard_normal((N, H, W, C)))
    a = Q(O.act_fwd(np.float32(np.float64(z) * np.float32(gamma * invstd) + np.float32(beta - mean * gamma * invstd)), O.ACT_HSWISH).astype(np.float64))
    y = ops.affine_act_bf16(TB(z), bn.scale, bn.shift, ops.ACT_HSWISH, residual=TB(r))
    close_bf16(y, Q(a + r), 'affine_act + residual')
    # pooling / SE multiply
    pooled = ops.global_avgpool_fwd_bf16(TB(z), bn.scale, bn.shift, ops.ACT_HSWISH)
    close_bf16(pooled, Q(a.mean(axis=(1, 2), keepdims=True)), 'global pooling')
    s = Q(rng.standard_normal((N, 1, 1, C)))
    sv = Q(O.act_fwd(s, O.ACT_HSIGMOID))
    ym = ops.scale_bcast_fwd_bf16(TB(z), TB(s), bn.scale, bn.shift, ops.ACT_HSWISH, ops.ACT_HSIGMOID)
    close_bf16(ym, Q(a * sv), 'SE multiply')
    gx, gs = ops.scale_bcast_bwd_bf16(TB(g), TB(z), TB(s), bn.scale, bn.shift, ops.ACT_HSWISH, ops.ACT_HSIGMOID)
    close_bf16(gx, Q(g * sv), 'SE multiply: gradient w.r.t. the tensor')
    close_bf16(gs, Q((g * a).sum(axis=(1, 2), keepdims=True)), 'SE multiply: gradient w.r.t. the scale', slack=2.0)
    # resize 17x19 -> 65x73 and back
    yr = ops.resize_bilinear_fwd_bf16(TB(z), 65, 73)
    close_bf16(yr, Q(O.resize_bilinear_fwd(z, 65, 73)), 'resize')
    gbig = Q(rng.standard_normal((N, 65, 73, C)))
    close_bf16(ops.resize_bilinear_bwd_bf16(TB(gbig), H, W), Q(O.resize_bilinear_bwd(gbig, H, W)), 'resize backward', slack=2.0)



@pytest.mark.parametrize('case', [(2, 17, 19, 256, 65, 73), (1, 9, 11, 20, 33, 41), (1, 16, 32, 256, 64, 128), (3, 5, 7, 24, 5, 7), (1, 1, 1, 64, 9, 5)])
def test_resize_bf16_channel_vectors(ops, case):
    """dl3p_resize_bilinear_{fwd,bwd}_bf16 with 8 channels per thread (16-byte accesses: channel counts that are multiples of 8) and
    with 4 (the rest); up-sampling by non-integer factors, the identity size, a 1 x 1 source"""
    N, h, w, C, H, W = case
    rng = np.random.default_rng(h * w + C)
    z = Q(rng.standard_normal((N, h, w, C)))
    close_bf16(ops.resize_bilinear_fwd_bf16(TB(z), H, W), Q(O.resize_bilinear_fwd(z, H, W)), 'resize')
    g = Q(rng.standard_normal((N, H, W, C)))
    close_bf16(ops.resize_bilinear_bwd_bf16(TB(g), h, w), Q(O.resize_bilinear_bwd(g, h, w)), 'resize backward', slack=2.0 + (H // h) * (W // w) / 8.0)


@pytest.mark.parametrize('case', [(2, 33, 33, 32, 3, 1, 1), (2, 17, 23, 64, 3, 2, 1), (1, 20, 20, 8, 1, 2, 1), (2, 19, 19, 16, 3, 1, 2),
                                  (1, 12, 30, 4, 7, 2, 1)])
def test_col2im_bf16(ops, case):
    """the data gradient of a dense k x k conv on the mixed path (the transposed im2col gather behind the GEMM): bf16 in, fp32 sums,
    one rounding at the store -- against the fp64 transpose of the oracle's im2col on the same bf16 values"""
    N, H, W, Cin, k, stride, rate = case
    rng = np.random.default_rng(N + H + Cin + k)
    Ho, Wo, pt, pl = ops.conv_geometry(H, W, k, stride, rate, 'same')
    kp = (k * k * Cin + 7) // 8 * 8
    gcol = Q(rng.standard_normal((N, Ho, Wo, kp)))
    gcol[..., k * k * Cin:] = 0
    # reference: scatter every (output pixel, tap) onto its input pixel
    want = np.zeros((N, H, W, Cin))
    for ky in range(k):
        for kx in range(k):
            for oy in range(Ho):
                iy = oy * stride - pt + ky * rate
                if iy < 0 or iy >= H:
                    continue
                for ox in range(Wo):
                    ix = ox * stride - pl + kx * rate
                    if 0 <= ix < W:
                        want[:, iy, ix, :] += gcol[:, oy, ox, (ky * k + kx) * Cin:(ky * k + kx + 1) * Cin]
    got = ops.col2im_bf16(TB(gcol), (N, H, W, Cin), k, stride, rate)
    close_bf16(got, Q(want), 'col2im')
    base = Q(rng.standard_normal((N, H, W, Cin)))
    got = ops.col2im_bf16(TB(gcol), (N, H, W, Cin), k, stride, rate, out=TB(base), accumulate=True)
    close_bf16(got, Q(want + base), 'col2im accumulate')


@pytest.mark.parametrize('case', [(2, 33, 33, 64, 3, 2, (1, 1, 1, 1)), (1, 16, 20, 8, 3, 2, (1, 1, 1, 1)), (2, 9, 9, 4, 2, 2, (0, 1, 0, 1)),
                                  (1, 257, 257, 64, 3, 2, (1, 1, 1, 1))])
def test_maxpool_bf16(ops, case):
    """ZeroPadding2D + MaxPooling2D on the mixed path (ResNet50's pool1 behind bn_conv1 + ReLU, deeplabv3p_resnet50.py:262-267):
    the prologue value is rounded to bf16 like every consumer-side prologue, the maximum of bf16 values is exact, the backward sums
    dy in fp32 over the windows a pixel won"""
    N, H, W, C, k, stride, pad = case
    rng = np.random.default_rng(H + C)
    z = Q(rng.standard_normal((N, H, W, C)))
    sc, sh = rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.3
    a = Q(np.maximum(z * sc + sh, 0.0))
    want, arg = O.maxpool2d_fwd(a, k, stride, pad)
    Ho, Wo = want.shape[1], want.shape[2]
    argmax = torch.zeros(N * Ho * Wo * C, dtype=torch.uint8, device=DEV)
    got = ops.maxpool2d_fwd_bf16(TB(z), k, stride, pad, TF(sc), TF(sh), ops.ACT_RELU, argmax=argmax)
    close_bf16(got, want, 'maxpool forward')
    frac = float((np64(got) == want).mean())
    assert frac > 0.999, frac                # (a prologue value at a rounding boundary may round the other way in fp32)
    gy = Q(rng.standard_normal(want.shape))
    # backward from the DEVICE's recorded winners (ties and boundary roundings then cannot differ)
    am = argmax.cpu().numpy().reshape(N, Ho, Wo, C).astype(np.int64)
    gx = np.zeros((N, H + pad[0] + pad[1], W + pad[2] + pad[3], C))
    n_i, c_i = np.meshgrid(np.arange(N), np.arange(C), indexing='ij')
    for oy in range(Ho):
        for ox in range(Wo):
            t = am[:, oy, ox, :]
            np.add.at(gx, (n_i, oy * stride + t // k, ox * stride + t % k, c_i), gy[:, oy, ox, :])
    gx = gx[:, pad[0]:pad[0] + H, pad[2]:pad[2] + W, :]
    got = ops.maxpool2d_bwd_bf16(TB(gy), argmax, (N, H, W, C), k, stride, pad)
    close_bf16(got, Q(gx), 'maxpool backward')
    base = Q(rng.standard_normal((N, H, W, C)))
    got = ops.maxpool2d_bwd_bf16(TB(gy), argmax, (N, H, W, C), k, stride, pad, out=TB(base), accumulate=True)
    close_bf16(got, Q(gx + base), 'maxpool backward accumulate')


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(1e-30, np.linalg.norm(a) * np.linalg.norm(b)))


@pytest.mark.parametrize('model_type,H,W', [('mobilenetv3large', 64, 96), ('mobilenetv3large', 128, 256), ('mobilenetv3large_lite', 65, 65),
                                            ('mobilenetv2', 65, 65), ('xception', 65, 65), ('resnet50', 65, 65)])
def test_train_step_bf16_matches_rounding_oracle(model_type, H, W):
    """Whole train step in bf16 against the fp64 oracle with bf16 rounding at the device's storage points, kept on the
    device's trajectory (oracle/np_net.py Net.force): two bf16 evaluations of one model drift apart after the first element
    that fp32 accumulation rounds the other way -- BatchNorm over the few hundred samples of these maps spreads it over
    the whole channel (scripts/bf16_layer_diff.py shows bit-identical layers up to that point, then exponential growth).
    Each conv / depthwise layer's output is therefore compared with what the oracle computes FROM THE DEVICE'S inputs to
    that layer, and the oracle then continues with the device's tensor:
      * forward: every one of the ~100 conv layers within one bf16 ulp (+ 2e-3 of the layer's range) on 99.9 % of its
        elements; logits and loss (fp32 head) to 1e-3;
      * backward: teacher-forced as well (Net.force_grad): the oracle's backward continues, at every conv output, with the
        gradient the DEVICE stored there.  Activation gradients are compared segment by segment (two bf16 ulps), parameter
        gradients become layer-local quantities and are held to 4e-3 of their scale per tensor, 2e-3 in relative L2 over all
        (round 2 accepted cos > 0.98 / relative L2 < 0.2 against an un-forced backward)."""
    from oracle.np_net import OracleModel
    pkg = load_pkg()
    mp = pkg.mixed_precision
    N, C = 2, 19
    mp.set_policy(mp.Policy('mixed_bfloat16'))
    try:
        m = pkg.get_deeplabv3p_model(model_type, C, (H, W), 16, training=True)
    finally:
        mp.set_policy(mp.Policy('float32'))
    assert m.bf16
    m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    o = OracleModel(model_type, C, (H, W), 16, dtype=np.float64, seed=0)
    o.net.bf16 = True
    rng = np.random.default_rng(42)
    for k, v in o.net.params.items():
        if k.endswith('/gamma'):
            v[...] = rng.uniform(0.5, 1.5, v.shape)
        elif k.endswith('/beta') or k.endswith('/bias'):
            v[...] = rng.standard_normal(v.shape) * 0.1
    m.set_weights_by_name(dict(o.net.params))
    m.use_graphs = False
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    loss = m.train_on_batch(x, y)
    ex = m._executor(N, True)
    assert ex.bf16 and ex.buf[m.graph.input.tensor.id].dtype == torch.bfloat16
    drop = [op for op in m.graph.ops if op.kind == 'materialize' and op.rate > 0][0]
    mask = ex.dropout_mask(drop).cpu().numpy()
    convs = [op for op in m.graph.ops if op.kind in ('conv_pw', 'conv_dense', 'conv_dw')]
    real = {op.name: op.layer.params[0].shape[-1] if op.kind != 'conv_dw' else op.c for op in convs}
    o.net.force = {op.name: ex.view(op.out).float().cpu().numpy()[..., :real[op.name]] for op in convs}
    o.net.record = {}
    # ... and the backward pass the same way: the device's d loss / d (conv output) of every layer (bf16 as stored; the
    # logits layer's is fp32) is what the oracle's backward continues with at that layer
    o.net.force_grad = {op.name: ex.view(op.out, grad=True).float().cpu().numpy()[..., :real[op.name]] for op in convs
                        if op.out.requires_grad}
    o.net.record_grad = {}
    o.net.grad_term_norm = {}
    _, ce, logits_ref = o.loss_and_grads(x, y, {'aspp_dropout': mask})
    assert set(o.net.record) == set(o.net.force)
    for op in convs:
        ref, got = o.net.record[op.name], o.net.force[op.name]
        tol = 1.01 * 2.0 ** -8 * np.abs(ref) + 2e-3 * np.abs(ref).max() + 1e-30
        if op.name == 'conv_upsample':
            tol = 1e-4 * max(1.0, np.abs(ref).max())          # the logits layer writes fp32
        frac = float((np.abs(got - ref) <= tol).mean())
        assert frac > 0.999, (op.name, frac, float(np.abs(got - ref).max()), float(np.abs(ref).max()))
    ops = load_pkg('ops')
    out = ops.upsample_softmax_ce(ex.view(m.head.tensor), C, H, W, want_logits=True)
    lg = out['logits'][..., :C].cpu().numpy()
    assert np.abs(lg - logits_ref).max() < 1e-3 * max(1.0, np.abs(logits_ref).max())
    assert abs(loss - ce) < 1e-3 * max(1.0, abs(ce)), (loss, ce)
    # backward, layer by layer.  (a) activation gradients: what the oracle derives for a conv output from
    # the DEVICE's gradients at the conv outputs downstream of it -- one data-gradient kernel, the BatchNorm backward with the
    # activation derivative, Add / concat / resize / pooling / squeeze-excite transposes in between, each storing bf16 -- to two
    # bf16 ulps of the element + 2^-7 of the tensor's rms on 99.9 % of the elements, 1 % in relative L2
    assert set(o.net.record_grad) == set(o.net.force_grad)
    report = []
    for op in convs:
        if op.name not in o.net.force_grad:
            continue
        ref, got = np.asarray(o.net.record_grad[op.name], np.float64), o.net.force_grad[op.name].astype(np.float64)
        rms = float(np.sqrt((ref ** 2).mean()))
        if rms < 1e-12:
            assert float(np.abs(got).max()) < 1e-9, op.name
            continue
        if op.name == 'conv_upsample':          # fp32 gradient of the logits (softmax - onehot through the resize transpose)
            tol = 1e-5 * np.abs(ref).max() + 0 * ref
        else:
            tol = 2.0 ** -6 * np.abs(ref) + 2.0 ** -7 * rms
        frac = float((np.abs(got - ref) <= tol).mean())
        rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        report.append((op.name, round(frac, 5), round(rel, 5)))
    # (b) parameter gradients (fp32 on the device): each is now a LAYER-LOCAL quantity -- the weight gradient of a conv from the
    # device's own input activations and output gradient, a BatchNorm's (dgamma, dbeta) from the device's gradients one segment
    # downstream -- at the tolerance of the op tests (1e-4 .. 2e-3 of the tensor's scale), not cos > 0.98
    st = m._store
    worst, num, den, bn_ratios = [], 0.0, 0.0, []
    for p in m.graph.all_params():
        ge = o.net.grads.get(p.name)
        if not p.trainable or ge is None:
            continue
        g = st.get(p, st.G).astype(np.float64)
        num += float(((g - ge) ** 2).sum()); den += float((ge ** 2).sum())
        if np.abs(ge).max() < 1e-7 and o.net.grad_term_norm.get(p.name) is None:
            assert np.abs(g).max() < 1e-5, p.name
            continue
        # A BatchNorm's (dgamma, dbeta) are sums of M bf16-stored terms whose rounding errors do not cancel the way the terms
        # do (the beta in front of a conv + BatchNorm pair has an exactly ZERO gradient -- the next BatchNorm removes the shift
        # -- so what any run computes for it is that noise and nothing else).  Each term was rounded to bf16 several times on its
        # way (every consumer's data gradient accumulates into the buffer in bf16, then the activation derivative): the error of
        # the sum is held against the l2 norm of the terms, 2^-9 per rounding and term, NOISE_ULPS roundings' worth.
        noise = o.net.grad_term_norm.get(p.name)
        if noise is not None and p.name.startswith('image_pooling'):
            continue        # 2 samples per channel at batch 2: xhat = +-1, the sums are differences of two numbers (degenerate)
        if noise is not None:
            per_ch = np.abs(g - ge) / np.maximum(noise.reshape(ge.shape), 1e-30)
            ratio = float(per_ch.max())
            frac_out = float((per_ch > NOISE_ULPS * 2.0 ** -9).mean())
            bn_ratios.append((p.name, round(ratio * 512, 2), round(frac_out, 4)))
            # (a pre-activation at rounding distance of a kink takes the other branch of the derivative in fp32 than in fp64:
            # an O(1) change of ONE of the channel's terms -- a channel in a hundred may hold one)
            if frac_out > 0.01 or ratio > 0.5:
                worst.append((p.name, 'bn-noise', round(ratio * 512, 2), round(frac_out, 4)))
            continue
        err = float(np.abs(g - ge).max() / np.abs(ge).max())
        rel = float(np.linalg.norm(g - ge) / np.linalg.norm(ge))
        # (image_pooling's kernel gradient is a sum over the batch's TWO pooled rows: a one-ulp difference in a pooled bf16 value
        # -- 2^-8 of it -- is 2^-8 of the element; the relative L2 over the tensor stays at 5e-4)
        if err > (1e-2 if p.name.startswith('image_pooling') else 4e-3) or rel > 4e-3:
            worst.append((p.name, round(err, 5), round(rel, 5)))
    _record_bf16_backward(dict(model=model_type, H=H, W=W, worst_activation_gradients=sorted(report, key=lambda r: -r[2])[:8],
                               lowest_fraction=sorted(report, key=lambda r: r[1])[:8],
                               overall_relative_l2=float(np.sqrt(num / den)), outside=worst[:10],
                               bn_sum_error_in_roundings=sorted(bn_ratios, key=lambda r: -r[1])[:8]))
    # (measured: worst layer 0.9961 of the elements within two ulps -- image_pooling, 38 values -- and 3.8e-3 in relative L2)
    # (image_pooling: the gradient arrives through a BatchNorm over the batch's 2 samples per channel -- xhat = +-1, its backward
    # is a difference of two nearly equal numbers; 3.8e-3 on the MobileNets, 7.2e-3 behind ResNet50's 2048-channel sum)
    bad_act = [r for r in report if not (r[1] > ACT_FRAC and r[2] < (2.5 * ACT_REL if r[0] == 'image_pooling' else ACT_REL))]
    assert not bad_act, bad_act[:10]
    assert not worst, worst[:10]
    assert np.sqrt(num / den) < 2e-3, np.sqrt(num / den)
    w = m.get_weights_by_name()
    assert all(v.dtype == np.float32 and np.isfinite(v).all() for v in w.values())


def test_bf16_full_size_properties():
    """BASELINE configs[4] per-GPU shape: MobileNetV3-Large, 1024 x 2048, 19 classes, batch 1, bf16.  Size-independent
    properties: deterministic replay (eager == hipGraph, bit for bit), loss ~ ln(19) at initialisation, finite weights"""
    pkg = load_pkg()
    mp = pkg.mixed_precision
    N, C, H, W = 1, 19, 1024, 2048
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)

    def run(use_graphs):
        mp.set_policy(mp.Policy('mixed_bfloat16'))
        try:
            m = pkg.get_deeplabv3p_model('mobilenetv3large', C, (H, W), 16, training=True)
        finally:
            mp.set_policy(mp.Policy('float32'))
        m.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
        m.use_graphs = use_graphs
        return m, [m.train_on_batch(x, y) for _ in range(3)]
    ma, la = run(False)
    wa = ma.get_weights_by_name()
    del ma
    torch.cuda.empty_cache()
    mb, lb = run(True)
    wb = mb.get_weights_by_name()
    assert la == lb, (la, lb)
    assert all(np.array_equal(wa[k], wb[k]) for k in wa)
    assert abs(la[0] - np.log(C)) < 0.5 and all(np.isfinite(v).all() for v in wa.values())


# ---------------------------------------------------------------------------------------------------------------------------
# round 4: the layer-local check AT 1024 x 2048.  The NumPy oracle needs many minutes for one configs[4]
# image, so the reference here is float64 torch ON THE DEVICE (matmul / shifted multiply-adds -- no kernel of this repo, no conv
# library), layer by layer from the DEVICE's own stored tensors, with bf16 rounding at the points the policy defines.
def _act64(u, act):
    O_ = load_pkg('ops')
    if act == O_.ACT_NONE:
        return u
    if act == O_.ACT_RELU:
        return u.clamp_min(0)
    if act == O_.ACT_RELU6:
        return u.clamp(0, 6)
    hs = (u + 3).clamp(0, 6) / 6
    return hs if act == O_.ACT_HSIGMOID else u * hs


def _taps(a, k, stride, rate, pad_t, pad_l, Ho, Wo):
    """the k x k shifted, strided views of a zero-padded (N, H, W, C) tensor: [(ky, kx, view (N, Ho, Wo, C))]"""
    N, H, W, C = a.shape
    need_h, need_w = (Ho - 1) * stride + (k - 1) * rate + 1, (Wo - 1) * stride + (k - 1) * rate + 1
    ap = torch.nn.functional.pad(a, (0, 0, pad_l, max(0, need_w - W - pad_l), pad_t, max(0, need_h - H - pad_t)))
    for ky in range(k):
        for kx in range(k):
            yield ky, kx, ap[:, ky * rate: ky * rate + (Ho - 1) * stride + 1: stride, kx * rate: kx * rate + (Wo - 1) * stride + 1: stride, :]


def test_bf16_every_layer_at_1024x2048_matches_float64_on_the_devices_own_inputs(monkeypatch):
    """BASELINE configs[4] at ITS size (MobileNetV3-Large + ASPP + decoder, 1024 x 2048, 19 classes, batch 1, bf16): every conv /
    depthwise / dense layer's FORWARD output and WEIGHT GRADIENT against float64 computed from the device's own input activations
    (rounded to bf16 where the policy rounds: the prologue's act(z scale + shift), the kernel mirror) and the device's own output
    gradient -- every bf16 kernel family at its production launch: wave-streaming and staged GEMMs, the few-row GEMMs behind the
    poolings, 3 x 3 window and 5 x 5 strip depthwise kernels incl. the atrous ones, the stem, their weight-gradient twins.
    Tolerances of the 128 x 256 teacher-forced test: outputs within one bf16 ulp (+ 2e-3 of the layer's range) on 99.9 % of the
    elements, kernel gradients to 4e-3 of their scale."""
    monkeypatch.setenv('DL3P_FOLD_APPLY', '0')
    monkeypatch.setenv('DL3P_GRAD_ALIAS', '0')
    pkg = load_pkg()
    O_ = load_pkg('ops')
    mp = pkg.mixed_precision
    N, C, H, W = 1, 19, 1024, 2048
    mp.set_policy(mp.Policy('mixed_bfloat16'))
    try:
        m = pkg.get_deeplabv3p_model('mobilenetv3large', C, (H, W), 16, training=True)
    finally:
        mp.set_policy(mp.Policy('float32'))
    m.compile(optimizer=pkg.SGD(0.0), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))      # lr 0: the weights stay what forward used
    m.use_graphs = False
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    loss = m.train_on_batch(x, y)
    assert np.isfinite(loss)
    ex = m._executor(N, True)
    assert ex.bf16
    weights = m.get_weights_by_name()
    st = m._store
    convs = [op for op in m.graph.ops if op.kind in ('conv_pw', 'conv_dense', 'conv_dw')]
    assert len(convs) > 60
    f64 = dict(dtype=torch.float64, device=DEV)
    bad, kinds = [], set()
    for op in convs:
        v = op.x
        cin = op.c if op.kind == 'conv_dw' else op.cin
        a = ex.view(v.tensor)[..., :cin].double()
        if v.group is not None:
            sc = ex.gscale[v.group.id][v.goff:v.goff + cin].double()
            sh = ex.gshift[v.group.id][v.goff:v.goff + cin].double()
            a = (a * sc + sh).float().double()               # the prologue's single fp32 fma
        if v.group is not None or v.act != O_.ACT_NONE:
            a = _act64(a, v.act).float().to(torch.bfloat16).double()      # ... rounded to bf16 as every Keras layer output is
        w = torch.from_numpy(weights[op.w.name]).to(DEV)
        wb = w.to(torch.bfloat16).double()                    # the bf16 mirror the kernels read
        cout = w.shape[-1] if op.kind != 'conv_dw' else op.c
        got = ex.view(op.out)[..., :cout].double()
        dz = ex.view(op.out, grad=True)[..., :cout].double() if op.out.requires_grad else None
        if op.kind == 'conv_pw':
            ref = (a.reshape(-1, cin) @ wb.reshape(cin, cout)).reshape(got.shape)
            gw = (a.reshape(-1, cin).t() @ dz.reshape(-1, cout)).reshape(w.shape) if dz is not None else None
        elif op.kind == 'conv_dw':
            ref = torch.zeros(got.shape, **f64)
            gw = torch.zeros(w.shape, **f64) if dz is not None else None
            for ky, kx, s in _taps(a, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo):
                ref += s * wb[ky, kx, :, 0]
                if gw is not None:
                    gw[ky, kx, :, 0] = (s * dz).sum((0, 1, 2))
        else:
            ref = torch.zeros(got.shape, **f64)
            gw = torch.zeros(w.shape, **f64) if dz is not None else None
            for ky, kx, s in _taps(a, op.k, op.stride, op.rate, op.pad_t, op.pad_l, op.Ho, op.Wo):
                ref += (s.reshape(-1, cin) @ wb[ky, kx]).reshape(got.shape)
                if gw is not None:
                    gw[ky, kx] = s.reshape(-1, cin).t() @ dz.reshape(-1, cout)
        if getattr(op, "b", None) is not None:
            ref = ref + torch.from_numpy(weights[op.b.name]).to(DEV).double()
        rmax = float(ref.abs().max())
        tol = 1.01 * 2.0 ** -8 * ref.abs() + 2e-3 * rmax + 1e-30
        if op.name == 'conv_upsample':
            tol = torch.full_like(ref, 1e-4 * max(1.0, rmax))      # the logits layer writes fp32
        frac = float(((got - ref).abs() <= tol).double().mean())
        rows = got.numel() // cout
        kinds.add((op.kind, op.k, op.stride, op.rate, 'long' if rows >= 65536 else ('few' if rows <= 64 else 'mid')))
        if frac <= 0.999:
            bad.append((op.name, 'forward', frac, float((got - ref).abs().max()), rmax))
        if gw is not None and op.w.trainable:
            g = torch.from_numpy(np.ascontiguousarray(st.get(op.w, st.G))).to(DEV).double().reshape(gw.shape)
            scale = float(gw.abs().max())
            if scale > 1e-12:
                err = float((g - gw).abs().max()) / scale
                rel = float(torch.linalg.norm(g - gw) / torch.linalg.norm(gw))
                if err > 4e-3 or rel > 4e-3:
                    bad.append((op.name, 'weight gradient', err, rel))
        del a, ref, got, dz, gw
    assert not bad, bad[:10]
    # the launch shapes this walked: long / mid / few-row pointwise, 3 x 3 and 5 x 5 depthwise at strides 1 / 2 and rates 1 / 2 / 6 / 12 / 18, the stem
    assert {k[0] for k in kinds} == {'conv_pw', 'conv_dense', 'conv_dw'}
    assert any(k[0] == 'conv_dw' and k[1] == 5 for k in kinds) and any(k[0] == 'conv_dw' and k[3] == 18 for k in kinds)
    assert any(k[0] == 'conv_pw' and k[4] == 'long' for k in kinds) and any(k[0] == 'conv_pw' and k[4] == 'few' for k in kinds)
