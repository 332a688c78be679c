"""Known answers the reference does pin (README.md:312-317 FLOPs/params table) for BOTH
the oracle graphs and the product graphs, the get_deeplabv3p_model() surface, and the C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import load_pkg, ROOT

# trainable / non-trainable parameter counts with 21 classes
KNOWN = {'mobilenetv2_lite': (2113557, 33088, 54), 'mobilenetv2': (2719813, 38784, 65),
         'xception': (41055413, 202800, 146), 'mobilenetv3large': (3514453, 28752, 59),
         # README.md:317 pins MobileNetV3Small Lite at 1.06 M parameters (its table counts the trainable ones: MobileNetV2
         # Lite 2.11 M = 2113557); the other two rows have no published figure and pin oracle == product only
         'mobilenetv3small_lite': (1057717, 12496, 36), 'mobilenetv3small': (1484165, 16848, 47),
         'mobilenetv3large_lite': (3036357, 24016, 48),
         'resnet50': (26722693, 70720, 67)}                  # README.md:314: 26.72 M


@pytest.mark.parametrize('mt', ['mobilenetv2_lite', 'mobilenetv2', 'mobilenetv3large', 'mobilenetv3small_lite',
                                'mobilenetv3small', 'mobilenetv3large_lite', 'resnet50'])
def test_oracle_param_counts(mt):
    from oracle.np_net import OracleModel
    m = OracleModel(mt, 21, (33, 33), 16)
    tr, ntr, nbn = KNOWN[mt]
    assert m.net.n_params(True) == tr
    assert m.net.n_params(False) == ntr
    if mt == 'mobilenetv3small_lite':
        assert round(tr / 1e6, 2) == 1.06               # README.md:317
    if mt == 'resnet50':
        assert round(tr / 1e6, 2) == 26.72              # README.md:314
    assert sum(1 for n in m.net.order if n.endswith('/gamma')) == nbn


# README.md:312-317, column FLOPS at 512 x 512, OS 16, 21 classes (TensorFlow's profiler: 2 flops per multiply-add of
# every convolution plus the elementwise ops, which weigh ~1 % on the large graphs and ~4 % on MobileNetV3Small-Lite)
README_GFLOPS = {'xception': 102.73, 'resnet50': 73.95, 'mobilenetv3large': 9.52, 'mobilenetv2_lite': 5.24,
                 'mobilenetv3small_lite': 1.36}


@pytest.mark.parametrize('mt', sorted(README_GFLOPS))
def test_forward_flops_match_readme(mt):
    """MAC-count known answer: the convolutions of the product graph at the README's geometry, 2 flops per
    multiply-add, reproduce the published figure from below (the remainder is the elementwise work)"""
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model(mt, 21, (512, 512), 16, training=False)
    macs = 0
    for op in m.graph.ops:
        if op.kind in ('conv_pw', 'conv_dense'):
            kh, kw, cin, cout = op.w.shape
            macs += op.Ho * op.Wo * kh * kw * cin * cout
        elif op.kind == 'conv_dw':
            macs += op.Ho * op.Wo * op.k * op.k * op.c
    g = 2.0 * macs / 1e9
    ref = README_GFLOPS[mt]
    assert g <= ref and g >= ref * (0.95 if mt == 'mobilenetv3small_lite' else 0.99), (mt, g, ref)


def _product_models():
    return sorted(load_pkg().deeplab_model_map.keys())


@pytest.mark.parametrize('mt', ['mobilenetv2_lite', 'mobilenetv2', 'xception', 'mobilenetv3large', 'mobilenetv3small_lite',
                                'mobilenetv3small', 'mobilenetv3large_lite', 'resnet50'])
def test_product_param_counts_and_names_match_oracle(mt):
    pkg = load_pkg()
    if mt not in pkg.deeplab_model_map:
        pytest.skip(mt + ' not built yet')
    m = pkg.get_deeplabv3p_model(mt, 21, (513, 513), 16)
    tr, ntr, nbn = KNOWN[mt]
    ps = m.graph.all_params()
    assert sum(p.size for p in ps if p.weight_trainable) == tr
    assert sum(p.size for p in ps if not p.weight_trainable) == ntr
    assert sum(1 for l in m.layers if l.kind == 'BatchNormalization') == nbn
    if mt == 'xception':
        return   # the oracle Xception takes ~20 s to initialise; names are covered by the GPU parity test
    from oracle.np_net import OracleModel
    o = OracleModel(mt, 21, (33, 33), 16)
    assert [p.name for p in ps] == o.net.order            # same Keras weight order
    for p in ps:
        assert p.shape == o.net.params[p.name].shape, p.name
    # 19 classes: subtract 2*257
    m19 = pkg.get_deeplabv3p_model(mt, 19, (513, 513), 16)
    assert m19.count_params() == tr + ntr - 514


def test_factory_surface():
    pkg = load_pkg()
    with pytest.raises(ValueError, match='This model type is not supported now'):
        pkg.get_deeplabv3p_model('resnet101', 21, (513, 513), 16)
    with pytest.raises(ValueError):
        pkg.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 7)
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 16, training=True)
    assert m.output_shape == (None, 513 * 513, 21) and m.input_shape == (None, 513, 513, 3)
    assert m.layers[-1].name == 'pred_mask' and m.get_layer('conv_upsample').count_params() == 256 * 21 + 21
    mi = pkg.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 16, training=False)
    assert mi.output_shape == (None, 513, 513, 21)
    with pytest.raises(ValueError, match='num_classes'):
        pkg.get_deeplabv3p_model('mobilenetv2', 254, (65, 65), 16)          # PNG label maps: < 254 classes (train.py:34)
    assert pkg.get_deeplabv3p_model('mobilenetv2_lite', 150, (65, 65), 16).output_shape == (None, 65 * 65, 150)
    # freeze levels (model.py:106-115)
    m1 = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (513, 513), 16, freeze_level=1)
    assert not m1.get_layer('expanded_conv_16_project').trainable and m1.get_layer('aspp0').trainable
    m2 = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (513, 513), 16, freeze_level=2)
    assert not m2.get_layer('concat_projection').trainable and m2.get_layer('conv_upsample').trainable
    lines = []
    m.summary(print_fn=lines.append)
    assert any('Total params: 2,758,597' in l for l in lines)
    # which batch variance feeds the moving averages is stated where a user looks
    assert any('moving variance: biased' in l for l in lines)
    mu = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (65, 65), 16, bn_moving_variance='unbiased')
    lines = []
    mu.summary(print_fn=lines.append)
    assert any('moving variance: unbiased' in l and '2.11.0' in l for l in lines)
    # the reference's import path works
    import deeplabv3p.model as shim
    assert shim.get_deeplabv3p_model is pkg.get_deeplabv3p_model


def test_no_cpu_fallback():
    """the product path must fail loudly without the HIP device"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (65, 65), 16)
    m.compile()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.train_on_batch(np.zeros((1, 65, 65, 3), np.float32), np.zeros((1, 65 * 65, 1), np.float32))
    ops = load_pkg('ops')
    with pytest.raises(ops.Dl3pError):
        ops.pwconv_fwd(torch.zeros((4, 8)), torch.zeros((8, 8)))


def test_weights_roundtrip(tmp_path):
    pkg = load_pkg()
    a = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (65, 65), 16, seed=1)
    b = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (65, 65), 16, seed=2)
    path = str(tmp_path / 'w.npz')
    a.save(path)
    b.load_weights(path)
    for x, y in zip(a.get_weights(), b.get_weights()):
        np.testing.assert_array_equal(x, y)
    c = pkg.get_deeplabv3p_model('mobilenetv2_lite', 21, (65, 65), 16, weights_path=path, seed=3)
    np.testing.assert_array_equal(c.get_weights()[0], a.get_weights()[0])


def test_c_abi_exports_every_declared_symbol():
    """libdl3p.so loads and exports exactly what include/dl3p.h declares (no compute without a GPU)"""
    libm = load_pkg('_lib')
    protos = libm.parse_header()
    assert len(protos) >= 30
    assert os.path.exists(libm.LIBPATH), 'build libdl3p.so first: python __graft_entry__.py'
    cdll = ctypes.CDLL(libm.LIBPATH)
    for name in protos:
        assert hasattr(cdll, name), name
    out = subprocess.run(['nm', '-D', '--defined-only', libm.LIBPATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if ' T dl3p_' in l}
    assert exported == set(protos), (exported ^ set(protos))
    L = libm.lib()
    assert L.version() == 100 and L.device_cus() == 256
    # argument validation happens before any launch -> callable without a device
    with pytest.raises(libm.Dl3pError, match='multiple of 4'):
        L.dwconv2d_fwd(16, 6, None, None, 0, 16, 16, 6, None, ctypes.byref(ctypes.c_int()), 1, 4, 4, 6, 3, 1, 1, 1, 1,
                       4, 4, None)


def test_dispatch_options_and_tile_table():
    """dl3p_set_option: the knobs the tile tuner (scripts/tune_gemm.py) and the small-shape tests move; the measured tile
    table is well-formed (host-side only, no GPU)"""
    libm = load_pkg('_lib')
    L = libm.lib()
    for name, v in ((b'gemm_nt', 4), (b'gemm_mi', 1), (b'gemm_per_cu', 3), (b'wgrad_tile', 2), (b'wgrad_per_cu', 4),
                    (b'gemm_tuned', 0), (b'pw_small_min_rows', 64), (b'dw_per_cu', 4), (b'dw_want', 384), (b'dw_maxth', 8),
                    (b'dw_tuned', 0)):
        assert L.set_option(name, v) == 0
    for name, v in ((b'gemm_nt', 0), (b'gemm_mi', 0), (b'gemm_per_cu', 0), (b'wgrad_tile', -1), (b'wgrad_per_cu', 0),
                    (b'gemm_tuned', 1), (b'pw_small_min_rows', -1), (b'dw_per_cu', 0), (b'dw_want', 0), (b'dw_maxth', 0),
                    (b'dw_tuned', 1)):
        assert L.set_option(name, v) == 0
    with pytest.raises(libm.Dl3pError, match='unknown option'):
        L.set_option(b'no_such_knob', 1)
    import re
    rows = re.findall(r'\{(-?\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}',
                      open(os.path.join(os.path.dirname(libm.LIBPATH), 'csrc', 'gemm_tuned.h')).read())
    assert len(rows) > 100
    seen = set()
    for role, M, K, N, nt, mi, pc in (tuple(int(v) for v in r) for r in rows):
        if role < 0:
            continue
        assert 0 <= role <= 4 and M > 64 and K > 0 and N > 0
        assert (role, M, K, N) not in seen, 'duplicate key'
        seen.add((role, M, K, N))
        if role == 4:
            assert 0 <= nt <= 3 and mi >= 1            # weight gradient: tile index, workgroups per CU
        else:
            assert 1 <= nt <= 8 and mi in (1, 2) and 0 <= pc <= 16


def test_mixed_precision_policy_is_checked_when_the_model_is_built():
    """train.py:37-46 applies the policy to every model type.  A model type whose training graph holds an op without a bf16 kernel
    is refused by get_deeplabv3p_model -- not at the first train step: the mechanism stays, the list is empty since
    the dense-conv data gradient and max pooling got their bf16 kernels -- and 'mixed_float16' says that it runs as bf16"""
    import warnings
    pkg = load_pkg()
    mp = pkg.mixed_precision
    model_mod = load_pkg('model')
    try:
        mp.set_policy(mp.Policy('mixed_bfloat16'))
        for mt in ('xception', 'resnet50'):
            assert pkg.get_deeplabv3p_model(mt, 21, (65, 65), 16, training=True).bf16
        model_mod._NO_BF16_TRAINING['mobilenetv2'] = 'test entry'
        try:
            with pytest.raises(ValueError, match='mixed_bfloat16 training is not built'):
                pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
        finally:
            del model_mod._NO_BF16_TRAINING['mobilenetv2']
        m = pkg.get_deeplabv3p_model('mobilenetv3large', 19, (64, 96), 16, training=True)
        assert m.bf16
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            mp.set_policy(mp.Policy('mixed_float16'))
        assert any('mixed_bfloat16' in str(x.message) for x in w)
    finally:
        mp.set_policy(mp.Policy('float32'))
    assert not pkg.get_deeplabv3p_model('xception', 21, (65, 65), 16, training=True).bf16
