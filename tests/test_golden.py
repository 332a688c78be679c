"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the oracle still reproduces them (guards the checker).  GPU: the HIP path matches them."""
import os

import numpy as np
import pytest

from conftest import load_pkg
from oracle import np_ops as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(G, name)))


def test_oracle_reproduces_op_fixtures():
    d = load('ops_v1.npz')
    a = O.act_fwd(d['dw_x'].astype(np.float64) * d['dw_scale'] + d['dw_shift'], O.ACT_RELU)
    for r in (1, 6, 12, 18):
        np.testing.assert_allclose(O.dwconv2d_fwd(a, d['dw_w'].astype(np.float64), 1, r, 'same'), d['dw_y_r%d' % r],
                                   rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.dwconv2d_fwd(d['dw_xe'].astype(np.float64), d['dw_w'].astype(np.float64), 2, 1, 'same'),
                               d['dw_ye_same_s2'], rtol=1e-5, atol=1e-5)
    big = O.resize_bilinear_fwd(d['head_z'][..., :21].astype(np.float64), 33, 33)
    loss, p, g = O.sparse_ce_fwd_bwd(big, d['head_labels'], 255)
    np.testing.assert_allclose(loss, d['head_loss'][0], rtol=1e-9)
    np.testing.assert_allclose(p, d['head_probs'], atol=1e-6)


def test_oracle_reproduces_model_fixture():
    from oracle.np_net import OracleModel
    d = load('mobilenetv2_lite_65.npz')
    o = OracleModel('mobilenetv2_lite', 21, (65, 65), 16, dtype=np.float64, seed=0)
    np.testing.assert_allclose(o.net.params['aspp0/kernel'], d['w:aspp0/kernel'], atol=1e-7)
    logits, probs = o.predict(d['x'])
    np.testing.assert_allclose(logits[:, ::4, ::4], d['logits_infer_s4'], atol=1e-5)
    total, ce, _ = o.loss_and_grads(d['x'], d['y'])
    np.testing.assert_allclose(ce, d['loss_train_nodropout'][0], rtol=1e-9)


@pytest.mark.gpu
def test_hip_ops_match_golden(ops):
    import torch
    d = load('ops_v1.npz')
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    for r in (1, 6, 12, 18):
        y = ops.dwconv2d_fwd(T(d['dw_x']), T(d['dw_w']), 1, r, 'same', T(d['dw_scale']), T(d['dw_shift']), ops.ACT_RELU)
        np.testing.assert_allclose(y.cpu().numpy(), d['dw_y_r%d' % r], rtol=1e-4, atol=1e-4)
        gx = ops.dwconv2d_bwd_data(T(d['dw_gy']), T(d['dw_w']), d['dw_x'].shape, 1, r, 'same')
        np.testing.assert_allclose(gx.cpu().numpy(), d['dw_gx_r%d' % r], rtol=1e-4, atol=1e-4)
        gw = ops.dwconv2d_bwd_weight(T(d['dw_x']), T(d['dw_gy']), 3, 1, r, 'same', T(d['dw_scale']), T(d['dw_shift']), ops.ACT_RELU)
        np.testing.assert_allclose(gw.cpu().numpy(), d['dw_gw_r%d' % r], rtol=2e-4, atol=2e-3)
    for pad, key in (('same', 'dw_ye_same_s2'), ((1, 1, 1, 1), 'dw_ye_pad11_s2')):
        y = ops.dwconv2d_fwd(T(d['dw_xe']), T(d['dw_w']), 2, 1, pad)
        np.testing.assert_allclose(y.cpu().numpy(), d[key], rtol=1e-4, atol=1e-4)
    y = ops.pwconv_fwd(T(d['pw_x']), T(d['pw_w']))
    np.testing.assert_allclose(y.cpu().numpy(), d['pw_y'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ops.pwconv_bwd_data(T(d['pw_gy']), T(d['pw_w'])).cpu().numpy(), d['pw_gx'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ops.pwconv_bwd_weight(T(d['pw_x']), T(d['pw_gy'])).cpu().numpy(), d['pw_gw'], rtol=1e-4, atol=1e-3)
    out = ops.upsample_softmax_ce(T(d['head_z']), 21, 33, 33, T(d['head_labels'].reshape(1, -1, 1)), 255, want_probs=True,
                                  want_logits=True, want_grad=True)
    np.testing.assert_allclose(out['logits'][..., :21].cpu().numpy(), d['head_logits'], atol=1e-5)
    np.testing.assert_allclose(out['probs'].cpu().numpy(), d['head_probs'], atol=1e-6)
    np.testing.assert_allclose(out['loss'].item(), d['head_loss'][0], rtol=1e-5)
    np.testing.assert_allclose(out['dlogits'][..., :21].cpu().numpy(), d['head_dlogits'], atol=1e-8, rtol=1e-4)
    np.testing.assert_allclose(ops.resize_bilinear_fwd(T(d['head_z']), 36, 36).cpu().numpy(), d['resize_y'], atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('mt', ['mobilenetv2_lite', 'mobilenetv2'])
def test_hip_model_matches_golden(mt):
    d = load(mt + '_65.npz')
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model(mt, 21, (65, 65), 16, training=False, seed=0)
    # the product initialises from the same seed with its own generator -> load the oracle's weights
    from oracle.np_net import OracleModel
    o = OracleModel(mt, 21, (65, 65), 16, dtype=np.float64, seed=0)
    np.testing.assert_allclose(o.net.params['conv_upsample/kernel'], d['w:conv_upsample/kernel'], atol=1e-7)
    m.set_weights_by_name(dict(o.net.params))
    p = m.predict(d['x'])
    assert np.abs(p[:, ::4, ::4] - d['probs_infer_s4']).max() < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# Reference-held pins: the label maps of /root/reference/example/*.png (tests/golden/make_reference_labels.py) with
# ANALYTIC answers that follow from deeplabv3p/loss.py:121-156 alone (masking, Keras' mean over all entries, the
# 1e-7 clip) -- not from this repo's oracle.
def _voc_cases():
    d = load('voc_example_labels.npz')
    C = 21
    for name in sorted(d):
        lab = d[name].astype(np.float32)
        valid = float((lab != 255).sum())
        yield name, lab, C, np.log(C) * valid / lab.size, -np.log(1.0 - 1e-7) * valid / lab.size


def test_oracle_ce_on_reference_labels():
    for name, lab, C, loss_uniform, loss_clipped in _voc_cases():
        H, W = lab.shape
        z = np.zeros((1, H, W, C))
        loss, p, g = O.sparse_ce_fwd_bwd(z, lab[None], 255)
        assert abs(loss - loss_uniform) < 1e-12, name
        want = np.full((H, W, C), 1.0 / C)
        iy, ix = np.nonzero(lab != 255)
        want[iy, ix, lab[iy, ix].astype(int)] -= 1.0
        want[lab == 255] = 0.0
        np.testing.assert_allclose(g[0], want / lab.size, atol=1e-15)
        onehot = np.zeros((1, H, W, C))
        onehot[0, iy, ix, lab[iy, ix].astype(int)] = 20.0
        loss, _, g = O.sparse_ce_fwd_bwd(onehot, lab[None], 255)
        assert abs(loss - loss_clipped) < 1e-15, name
        assert not g.any()                      # clipped probabilities have zero gradient (tf.clip_by_value)


@pytest.mark.gpu
def test_hip_head_on_reference_labels(ops):
    import torch
    for name, lab, C, loss_uniform, loss_clipped in _voc_cases():
        H, W = lab.shape
        labels = torch.from_numpy(lab[None].copy()).cuda()
        # all-equal small logits, upsampled 4x by the head itself
        z = torch.full((1, (H + 3) // 4, (W + 3) // 4, 24), 0.25, device='cuda')
        z[..., C:] = 0
        out = ops.upsample_softmax_ce(z, C, H, W, labels=labels, want_grad=True)
        assert abs(float(out['loss'].item()) - loss_uniform) < 2e-6 * loss_uniform, name
        g = out['dlogits'][0, ..., :C].cpu().numpy()
        iy, ix = np.nonzero(lab != 255)
        want = np.full((H, W, C), 1.0 / C)
        want[iy, ix, lab[iy, ix].astype(int)] -= 1.0
        want[lab == 255] = 0.0
        np.testing.assert_allclose(g, want / lab.size, atol=1e-6 / lab.size)
        # logits = 20 * onehot(label) at full resolution (identity resize): every labelled pixel is clipped
        zz = torch.zeros((1, H, W, 24), device='cuda')
        idx = torch.from_numpy(np.where(lab == 255, 0, lab).astype(np.int64)).cuda()
        zz[0].scatter_(2, idx[..., None], 20.0)
        out = ops.upsample_softmax_ce(zz, C, H, W, labels=labels, want_grad=True)
        # in float32 (the reference's dtype) the clip bound 1 - 1e-7 is the float 1 - 2^-23
        clipped32 = loss_clipped * np.log(np.float64(np.float32(1.0 - 1e-7))) / np.log(1.0 - 1e-7)
        assert abs(float(out['loss'].item()) - clipped32) < 1e-2 * clipped32, name
        assert not out['dlogits'].any()
