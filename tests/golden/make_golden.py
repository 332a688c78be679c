"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference (tf.keras, TensorFlow 2.11) cannot be imported in this environment, so these vectors
come from the fp64 CPU oracle (oracle/), whose ops are triangulated against torch CPU in
tests/test_oracle_ops.py -- PARITY UNPINNED against TensorFlow itself.
A fixture is data only: seeded inputs and the oracle's outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_ops as O          # noqa: E402
from oracle.np_net import OracleModel   # noqa: E402


def ops_fixture():
    rng = np.random.default_rng(2026)
    d = {}
    # atrous depthwise at the true ASPP geometry (33x33, rates 6/12/18), reduced channels
    x = rng.standard_normal((2, 33, 33, 8)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 8)) * 0.3).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    sh = (rng.standard_normal(8) * 0.3).astype(np.float32)
    gy = rng.standard_normal((2, 33, 33, 8)).astype(np.float32)
    d.update(dw_x=x, dw_w=w, dw_scale=sc, dw_shift=sh, dw_gy=gy)
    a = O.act_fwd(x.astype(np.float64) * sc + sh, O.ACT_RELU)
    for r in (1, 6, 12, 18):
        d['dw_y_r%d' % r] = O.dwconv2d_fwd(a, w.astype(np.float64), 1, r, 'same')
        gx, gw = O.dwconv2d_bwd(a, w.astype(np.float64), gy.astype(np.float64), 1, r, 'same')
        d['dw_gx_r%d' % r], d['dw_gw_r%d' % r] = gx, gw
    # stride 2, SAME at an even size (extra pad bottom/right) and Xception-style explicit pad
    xe = rng.standard_normal((1, 16, 20, 8)).astype(np.float32)
    d['dw_xe'] = xe
    d['dw_ye_same_s2'] = O.dwconv2d_fwd(xe.astype(np.float64), w.astype(np.float64), 2, 1, 'same')
    d['dw_ye_pad11_s2'] = O.dwconv2d_fwd(xe.astype(np.float64), w.astype(np.float64), 2, 1, (1, 1, 1, 1))
    # pointwise
    px = rng.standard_normal((300, 24)).astype(np.float32)
    pw = (rng.standard_normal((24, 48)) / 5).astype(np.float32)
    pg = rng.standard_normal((300, 48)).astype(np.float32)
    d.update(pw_x=px, pw_w=pw, pw_gy=pg, pw_y=px.astype(np.float64) @ pw, pw_gx=pg.astype(np.float64) @ pw.T,
             pw_gw=px.astype(np.float64).T @ pg)
    # batch norm (train) forward/backward
    z = (rng.standard_normal((2, 9, 9, 16)) * 2 + 0.5).astype(np.float32)
    g, b = rng.uniform(0.5, 1.5, 16).astype(np.float32), (rng.standard_normal(16) * 0.2).astype(np.float32)
    gz = rng.standard_normal(z.shape).astype(np.float32)
    y, cache, (bm, bv) = O.bn_train_fwd(z.astype(np.float64), g, b, 1e-3)
    gx, gg, gb = O.bn_train_bwd(O.act_bwd(y, gz.astype(np.float64), O.ACT_RELU6), cache)
    d.update(bn_z=z, bn_gamma=g, bn_beta=b, bn_g=gz, bn_y=y, bn_mean=bm, bn_var=bv, bn_dz=gx, bn_dgamma=gg,
             bn_dbeta=gb)
    # bilinear 33 -> 129 and the head (pred_resize + softmax + CE with ignore 255)
    rx = rng.standard_normal((1, 9, 9, 24)).astype(np.float32)
    rx[..., 21:] = 0
    lab = rng.integers(0, 21, (1, 33, 33)).astype(np.float32)
    lab[0, :3, :5] = 255
    big = O.resize_bilinear_fwd(rx[..., :21].astype(np.float64), 33, 33)
    loss, p, gl = O.sparse_ce_fwd_bwd(big, lab, 255)
    d.update(head_z=rx, head_labels=lab, head_logits=big, head_probs=p, head_loss=np.array([loss]), head_dlogits=gl,
             resize_y=O.resize_bilinear_fwd(rx.astype(np.float64), 36, 36))
    return d


def model_fixture(model_type, H, W):
    rng = np.random.default_rng(7)
    N, C = 2, 21
    o = OracleModel(model_type, C, (H, W), 16, dtype=np.float64, seed=0)
    x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    logits, probs = o.predict(x)
    # outputs are stored on a stride-4 pixel grid to keep the fixture small
    d = {'x': x, 'y': y, 'logits_infer_s4': logits[:, ::4, ::4].astype(np.float32),
         'probs_infer_s4': probs[:, ::4, ::4].astype(np.float32)}
    # weights are regenerated from the seed by both sides; store a few as a guard
    for k in ('Conv/kernel', 'conv_upsample/kernel', 'aspp0/kernel'):
        d['w:' + k] = o.net.params[k].astype(np.float32)
    total, ce, lg = o.loss_and_grads(x, y)            # no dropout mask injected -> dropout off
    d['loss_train_nodropout'] = np.array([ce])
    d['reg_loss'] = np.array([total - ce])
    for k in ('conv_upsample/kernel', 'conv_upsample/bias', 'concat_projection_BN/gamma'):
        d['g:' + k] = o.net.grads[k]
    return d


def f32(d):
    return {k: (v.astype(np.float32) if v.dtype == np.float64 and v.size > 64 else v) for k, v in d.items()}


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'ops_v1.npz'), **f32(ops_fixture()))
    np.savez_compressed(os.path.join(HERE, 'mobilenetv2_lite_65.npz'), **f32(model_fixture('mobilenetv2_lite', 65, 65)))
    np.savez_compressed(os.path.join(HERE, 'mobilenetv2_65.npz'), **f32(model_fixture('mobilenetv2', 65, 65)))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
