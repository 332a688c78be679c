"""The fused inverted-residual kernels (csrc/irb_fwd.hip, irb_bwd.hip: expand 1x1 -> BatchNorm -> ReLU6 -> depthwise 3x3 of
deeplabv3p_mobilenetv2.py:43-60 without the expanded tensor in HBM) against float64 torch on the same inputs, and against the
library's own unfused kernels."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def relerr(got, want):
    got, want = got.double(), want.double()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def sumerr(got, net, l1):
    """error of a float32-accumulated sum against float64, relative to the sum of the |terms| (what fp32 accumulation is bounded
    by: a sum over 10^6 pixels of zero-mean terms cancels by three orders of magnitude and its error does not)"""
    return float((got.double() - net.double()).abs().max() / l1.double().abs().max().clamp_min(1e-30))


def sum_ok(got, net, l1, net_tol, slack=None):
    """slack: what the elements whose ReLU6 branch is undecided at float32 resolution can move the sum by (see `amb` below)"""
    if relerr(got, net) < net_tol:
        return True
    err = (got.double() - net.double()).abs()
    if slack is not None:
        err = (err - slack.double()).clamp_min(0.0)
    return float(err.max() / l1.double().abs().max().clamp_min(1e-30)) < 2e-6


def tf_pad(H, k, s):
    out = -(-H // s)
    total = max((out - 1) * s + k - H, 0)
    return out, total // 2, total - total // 2


def dw64(a, w, stride):
    """a (N,H,W,C) float64, w (3,3,C) -> TF 'SAME' depthwise conv, NHWC"""
    N, H, W, C = a.shape
    _, pt, pb = tf_pad(H, 3, stride)
    _, pl, pr = tf_pad(W, 3, stride)
    x = F.pad(a.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(x, w.permute(2, 0, 1).unsqueeze(1), stride=stride, groups=C)
    return y.permute(0, 2, 3, 1)


def make(case, seed=0, big_mean=False):
    N, H, W, K, C, stride = case
    rng = np.random.default_rng(seed + H * 7 + K)
    d = dict(N=N, H=H, W=W, K=K, C=C, stride=stride)
    d['x'] = T(rng.standard_normal((N, H, W, K)))
    d['xs'] = T(rng.uniform(0.5, 1.5, K)); d['xh'] = T(rng.standard_normal(K) * (5.0 if big_mean else 0.3))
    d['w1'] = T(rng.standard_normal((K, C)) / np.sqrt(K))
    d['wdw'] = T(rng.standard_normal((3, 3, C)) * 0.4)
    d['gamma'] = T(rng.uniform(0.5, 1.5, C)); d['beta'] = T(rng.standard_normal(C) * 0.5 + 1.0)
    Ho, Wo = -(-H // stride), -(-W // stride)
    d['dy'] = T(rng.standard_normal((N, Ho, Wo, C)))
    return d


def ref64(d, bn_consts=None):
    """float64 chain with autograd; bn_consts = (scale, shift, mean, invstd) to use the DEVICE's coefficients"""
    x = d['x'].double()
    xh = (x * d['xs'].double() + d['xh'].double()).requires_grad_(True)
    w1 = d['w1'].double().requires_grad_(True)
    wdw = d['wdw'].double().requires_grad_(True)
    z1 = xh @ w1
    return x, xh, w1, wdw, z1


CASES = [(2, 33, 33, 16, 96, 2), (2, 32, 40, 16, 96, 2), (1, 65, 65, 24, 144, 1), (2, 31, 29, 24, 144, 2),
         (1, 17, 19, 32, 192, 1), (1, 20, 24, 32, 192, 2), (2, 30, 47, 16, 96, 1), (1, 129, 129, 16, 96, 2),
         # the launches of the timed step (BASELINE configs[1], batch 16: expanded_conv_1 and _3) and the batch-8 form of the second
         (16, 257, 257, 16, 96, 2), (16, 129, 129, 24, 144, 2), (8, 129, 129, 24, 144, 2)]


def test_lane_shift_primitives(ops):
    out = torch.zeros(128, device=DEV)
    ops.lib().irb_selftest(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    o = out.cpu().numpy()
    lane = np.arange(64)
    nxt = np.where(lane % 16 == 15, lane, lane + 1)
    prv = np.where(lane % 16 == 0, lane, lane - 1)
    # the row ends return SOME value of the row (never used by the kernels): only the interior is pinned
    inner_n, inner_p = lane % 16 != 15, lane % 16 != 0
    assert (o[:64][inner_n] == nxt[inner_n]).all(), o[:64]
    assert (o[64:][inner_p] == prv[inner_p]).all(), o[64:]


@pytest.mark.parametrize('K,C,M', [(16, 96, 5000), (24, 144, 3333), (32, 192, 70001)])
@pytest.mark.parametrize('big_mean', [False, True])
def test_covariance_statistics_equal_the_materialised_ones(ops, K, C, M, big_mean):
    """mean / variance of z = x W from sum x, sum x x^T == the statistics of the materialised z (float64), also for channels
    whose |mean| dwarfs their sigma (1e-6 relative)"""
    rng = np.random.default_rng(K + M)
    x = T(rng.standard_normal((M, K)))
    xs = T(rng.uniform(0.5, 1.5, K))
    xh = T(rng.standard_normal(K) * 0.3 + (30.0 if big_mean else 0.0))
    w1 = T(rng.standard_normal((K, C)) / np.sqrt(K))
    if big_mean:
        w1[:, 0] = 1.0                      # channel 0: mean ~ sum of the shifts, far above its sigma
    sums = ops.irb_cov_sums(x, xs, xh, ops.ACT_NONE)
    # what the kernel's fmaf gives: the product is exact in float64, so this is one rounding (up to 2^-29 double-rounding cases)
    xk = (x.double() * xs.double() + xh.double()).float().double()
    want = torch.cat([xk.sum(0), (xk.t() @ xk).reshape(-1)])
    assert relerr(sums, want) < 1e-9, relerr(sums, want)
    bn = ops.BNState(C, DEV)
    bn.gamma.copy_(T(rng.uniform(0.5, 1.5, C))); bn.beta.copy_(T(rng.standard_normal(C)))
    mm0, mv0 = bn.moving_mean.clone(), bn.moving_var.clone()
    ops.irb_bn_finalize_cov(bn, sums, w1, M)
    z = xk @ w1.double()
    mean, var = z.mean(0), z.var(0, unbiased=False)
    assert float(((bn.mean.double() - mean).abs() / (mean.abs() + var.sqrt())).max()) < 1e-6
    inv = 1.0 / torch.sqrt(var + bn.eps)
    assert relerr(bn.invstd, inv) < 1e-6, relerr(bn.invstd, inv)
    if big_mean:
        assert float(mean[0].abs() / var[0].sqrt()) > 20
    sc = bn.gamma.double() * inv
    assert relerr(bn.scale, sc) < 1e-6
    assert float((bn.shift.double() - (bn.beta.double() - mean * sc)).abs().max() / (mean.abs() * sc.abs() + 1).max()) < 1e-6
    # (the moving averages are float32 arithmetic on float32 state, as in dl3p_bn_finalize: 1 - 0.99 alone is 4e-7 off)
    assert relerr(bn.moving_mean, mm0.double() * bn.momentum + mean * (1 - bn.momentum)) < 1e-5
    assert relerr(bn.moving_var, mv0.double() * bn.momentum + var * (1 - bn.momentum)) < 1e-5


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('ct', [0, 1])
def test_fused_forward(ops, case, ct):
    d = make(case)
    N, H, W, K, C, stride = case
    assert ops.irb_supported((N, H, W, K), C, stride)
    ops.lib().irb_set_plan(ct, 0)
    try:
        x, xh, w1, wdw, z1 = ref64(d)
        # BatchNorm coefficients from the covariance route
        bn = ops.BNState(C, DEV)
        bn.gamma.copy_(d['gamma']); bn.beta.copy_(d['beta'])
        sums = ops.irb_cov_sums(d['x'], d['xs'], d['xh'], ops.ACT_NONE)
        ops.irb_bn_finalize_cov(bn, sums, d['w1'], N * H * W)
        part = ops.new_partials(C, DEV)
        y, rows = ops.irb_fwd(d['x'], d['w1'], bn.scale, bn.shift, ops.ACT_RELU6, d['wdw'], stride, in_scale=d['xs'],
                              in_shift=d['xh'], partials=part)
        a1 = torch.clamp(z1 * bn.scale.double() + bn.shift.double(), 0, 6)
        want = dw64(a1, wdw, stride).detach()
        assert y.shape == want.shape
        assert relerr(y, want) < 2e-6, relerr(y, want)
        p = part[:rows * 2 * C].view(rows, 2, C).double().sum(0)
        assert sum_ok(p[0], want.sum((0, 1, 2)), want.abs().sum((0, 1, 2)), 2e-5) and relerr(p[1], (want ** 2).sum((0, 1, 2))) < 2e-5
        # ... and the library's own unfused kernels on the same coefficients
        z1d = ops.pwconv_fwd(d['x'].view(-1, K), d['w1'], in_scale=d['xs'], in_shift=d['xh']).view(N, H, W, C)
        yu = ops.dwconv2d_fwd(z1d, d['wdw'], stride, in_scale=bn.scale, in_shift=bn.shift, in_act=ops.ACT_RELU6)
        assert relerr(y, yu) < 2e-6, relerr(y, yu)
    finally:
        ops.lib().irb_set_plan(0, 0)


def test_fused_forward_into_a_slice_and_from_a_slice(ops):
    case = (2, 33, 31, 16, 96, 2)
    d = make(case)
    N, H, W, K, C, stride = case
    wide_in = torch.zeros((N, H, W, K + 8), device=DEV); wide_in[..., 4:4 + K] = d['x']
    Ho, Wo = -(-H // 2), -(-W // 2)
    wide_out = torch.zeros((N, Ho, Wo, C + 12), device=DEV)
    sc = T(np.random.default_rng(1).uniform(0.5, 1.5, C)); sh = T(np.random.default_rng(2).standard_normal(C))
    ops.irb_fwd(wide_in[..., 4:4 + K], d['w1'], sc, sh, ops.ACT_RELU6, d['wdw'], stride, out=wide_out[..., 8:8 + C])
    y = ops.irb_fwd(d['x'], d['w1'], sc, sh, ops.ACT_RELU6, d['wdw'], stride)
    assert torch.equal(wide_out[..., 8:8 + C], y)
    assert float(wide_out[..., :8].abs().max()) == 0 and float(wide_out[..., 8 + C:].abs().max()) == 0


@pytest.mark.parametrize('case', CASES)
def test_fused_backward(ops, case):
    d = make(case, seed=3)
    N, H, W, K, C, stride = case
    if not ops.irb_supported((N, H, W, K), C, stride, backward=True):
        pytest.skip('backward serves C = 6K only')
    rng = np.random.default_rng(C)
    x, xh, w1, wdw, z1 = ref64(d)
    bn = ops.BNState(C, DEV)
    bn.gamma.copy_(d['gamma']); bn.beta.copy_(d['beta'])
    sums = ops.irb_cov_sums(d['x'], d['xs'], d['xh'], ops.ACT_NONE)
    ops.irb_bn_finalize_cov(bn, sums, d['w1'], N * H * W)
    sc, sh, mu, inv = bn.scale.double(), bn.shift.double(), bn.mean.double(), bn.invstd.double()
    u = z1 * sc + sh
    a1 = torch.clamp(u, 0, 6)
    a1.retain_grad()
    z2 = dw64(a1, wdw, stride)
    dy = d['dy'].double()
    (z2 * dy).sum().backward(retain_graph=True)
    da1 = a1.grad
    # the sum of the |terms| of the depthwise kernel's gradient (a1 >= 0 behind ReLU6): the same contraction on |dy|
    wabs = wdw.detach().clone().requires_grad_(True)
    (dw64(a1.detach(), wabs, stride) * dy.abs()).sum().backward()
    gwdw_l1 = wabs.grad
    gmask = ((u > 0) & (u < 6)).double()
    gp = da1 * gmask
    xhat1 = (z1.detach() - mu) * inv
    # elements within float32 rounding of a ReLU6 corner: the device may take the other branch there (at 10^8 elements a few per
    # channel do), which moves a sum by up to |da1| each -- the bounds below allow exactly that much on top of the rounding bound
    amb = (((u.detach().abs() < 1e-5) | ((u.detach() - 6).abs() < 1e-5))).double()
    flip = da1.abs() * amb
    # ---- pass A
    gwdw, part, rows = ops.irb_bwd_sums(d['x'], d['w1'], bn, ops.ACT_RELU6, d['wdw'], d['dy'], stride, in_scale=d['xs'],
                                        in_shift=d['xh'])
    assert sum_ok(gwdw, wdw.grad, gwdw_l1, 2e-5), (relerr(gwdw, wdw.grad), sumerr(gwdw, wdw.grad, gwdw_l1))
    p = part[:rows * 2 * C].view(rows, 2, C).double().sum(0)
    s1, s2 = gp.sum((0, 1, 2)), (gp * xhat1).sum((0, 1, 2))
    l1, l2 = gp.abs().sum((0, 1, 2)), (gp * xhat1).abs().sum((0, 1, 2))
    f1, f2 = flip.sum((0, 1, 2)), (flip * xhat1.abs()).sum((0, 1, 2))
    assert sum_ok(p[0], s1, l1, 2e-5, f1) and sum_ok(p[1], s2, l2, 2e-5, f2), (relerr(p[0], s1), relerr(p[1], s2),
                                                                            sumerr(p[0], s1, l1), sumerr(p[1], s2, l2))
    # ---- BatchNorm-backward coefficients (the library's own finalize), then pass B
    M = N * H * W
    ops.lib().bn_bwd_finalize(part.data_ptr(), rows, None, C, float(M), bn.gamma.data_ptr(), bn.invstd.data_ptr(),
                              bn.scale.data_ptr(), 0, bn.dgamma.data_ptr(), bn.dbeta.data_ptr(), bn.coef.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    assert sum_ok(bn.dgamma, s2, l2, 2e-5, f2) and sum_ok(bn.dbeta, s1, l1, 2e-5, f1)
    c0, c1, c2 = bn.coef.double().view(3, C)
    dz1 = c0 * (gp - c1 - xhat1 * c2)
    gw1_ref = xh.detach().reshape(-1, K).t() @ dz1.reshape(-1, C)
    gw1_l1 = xh.detach().abs().reshape(-1, K).t() @ dz1.abs().reshape(-1, C)
    gx_ref = dz1 @ w1.detach().t()
    dflip = flip * c0.abs()                                   # what an undecided branch moves dz1 by
    gw1_slack = xh.detach().abs().reshape(-1, K).t() @ dflip.reshape(-1, C)
    gx_slack = dflip @ w1.detach().abs().t()                  # ... and the input gradient of that pixel
    clean = (amb.sum(-1, keepdim=True) == 0).double()         # pixels with every branch decided
    # a BatchNorm in front of the block (z0 = the block's raw input here, as for expanded_conv_1)
    mu0 = T(rng.standard_normal(K) * 0.2); is0 = T(rng.uniform(0.5, 2.0, K))
    base = T(rng.standard_normal((N, H, W, K)))
    for accumulate in (False, True):
        gw1, gx, part0, rows0 = ops.irb_bwd_data(d['x'], d['w1'], bn, ops.ACT_RELU6, d['wdw'], d['dy'], stride, in_scale=d['xs'],
                                                 in_shift=d['xh'], out=base.clone() if accumulate else None, accumulate=accumulate,
                                                 front=(d['x'], d['xs'], d['xh'], ops.ACT_NONE, mu0, is0))
        assert sum_ok(gw1, gw1_ref, gw1_l1, 3e-5, gw1_slack), (relerr(gw1, gw1_ref), sumerr(gw1, gw1_ref, gw1_l1))
        tot = gx_ref + (base.double() if accumulate else 0.0)
        assert relerr(gx * clean, tot * clean) < 1e-5, relerr(gx * clean, tot * clean)
        assert float(((gx.double() - tot).abs() - gx_slack).max() / tot.abs().max()) < 1e-5
        p0 = part0[:rows0 * 2 * K].view(rows0, 2, K).double().sum(0)
        xhat0 = (x - mu0.double()) * is0.double()
        # (sum over pixels of a BatchNorm-backward result is 0 by construction: the first row is measured against sum |g|)
        e0 = float(((p0[0] - tot.sum((0, 1, 2))).abs() - gx_slack.sum((0, 1, 2))).max() / tot.abs().sum((0, 1, 2)).max())
        e1 = float(((p0[1] - (tot * xhat0).sum((0, 1, 2))).abs() - (gx_slack * xhat0.abs()).sum((0, 1, 2))).max()
                   / (tot * xhat0).abs().sum((0, 1, 2)).max())
        assert e0 < 1e-6 and e1 < 1e-6, (accumulate, e0, e1)
    # without a gradient for the input: the kernel gradient alone, same bits
    gw1b, none, _, _ = ops.irb_bwd_data(d['x'], d['w1'], bn, ops.ACT_RELU6, d['wdw'], d['dy'], stride, in_scale=d['xs'],
                                        in_shift=d['xh'], want_gx=False)
    assert none is None and torch.equal(gw1b, gw1)


def test_unsupported_shapes_are_refused(ops):
    assert not ops.irb_supported((2, 33, 33, 20), 120, 1)
    assert not ops.irb_supported((2, 33, 33, 16), 100, 1)
    assert ops.irb_supported((2, 33, 33, 16), 64, 1) and not ops.irb_supported((2, 33, 33, 16), 64, 1, backward=True)
    x = torch.zeros((2, 33, 33, 20), device=DEV)
    with pytest.raises(ops.Dl3pError):
        ops.irb_fwd(x, torch.zeros((20, 96), device=DEV), torch.ones(96, device=DEV), torch.zeros(96, device=DEV), 2,
                    torch.zeros((3, 3, 96), device=DEV))
