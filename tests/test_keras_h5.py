"""Keras HDF5 weight files: h5io.py (ctypes on libhdf5) against golden files written by
the real h5py with Keras' own call sequence (tests/golden/make_keras_h5.py), a round trip through the model facade, and
-- where an interpreter with h5py exists (this container: /opt/conda/bin/python3.9) -- h5py reading what h5io wrote."""
import importlib.util
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from conftest import load_pkg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
H5PY_PYTHON = '/opt/conda/bin/python3.9'


@pytest.fixture(scope='module')
def h5io():
    try:
        m = load_pkg('h5io')
        m.lib()
    except ImportError as e:
        pytest.skip(str(e))
    return m


def _expected(name, shape):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return rng.standard_normal(shape).astype(np.float32)


def _gen():
    spec = importlib.util.spec_from_file_location('make_keras_h5_spec', os.path.join(GOLD, 'make_keras_h5.py'))
    src = open(spec.origin).read().replace('import h5py\n', '')     # the layer tables only; h5py is not importable here
    ns = {}
    exec(compile(src, spec.origin, 'exec'), ns)
    return ns


@pytest.mark.parametrize('fname,whole', [('keras_weights_small.h5', False), ('keras_model_small.h5', True)])
def test_reads_h5py_written_keras_files(h5io, fname, whole):
    layers, attrs = h5io.read_keras_h5(os.path.join(GOLD, fname))
    small = _gen()['SMALL']
    assert [n for n, _ in layers] == [n for n, _ in small]
    for (n, ws), (_, spec) in zip(layers, small):
        assert [w for w, _ in ws] == [w for w, _ in spec]
        for (wn, arr), (_, shape) in zip(ws, spec):
            assert arr.dtype == np.float32 and arr.shape == tuple(shape)
            assert np.array_equal(arr, _expected(wn, shape)), wn
    assert attrs['backend'] == 'tensorflow' and attrs['keras_version'] == '2.11.0'
    if whole:
        assert json.loads(attrs['model_config'])['class_name'] == 'Functional'


def test_reads_chunked_name_attributes(h5io):
    """layer_names split into layer_names0, layer_names1 (numpy 'S' arrays, fixed-length strings)"""
    layers, _ = h5io.read_keras_h5(os.path.join(GOLD, 'keras_weights_chunked_names.h5'))
    many = _gen()['MANY']
    assert [n for n, _ in layers] == [n for n, _ in many]
    assert np.array_equal(layers[37][1][0][1], _expected('w37/gamma:0', (2,)))


def test_write_read_round_trip(h5io, tmp_path):
    rng = np.random.default_rng(0)
    layers = [('in', []), ('a', [('a/kernel:0', rng.standard_normal((3, 3, 4, 8)).astype(np.float32)),
                                 ('a/bias:0', rng.standard_normal(8).astype(np.float32))]),
              ('relu', []), ('b_BN', [('b_BN/gamma:0', np.ones(8, np.float32)), ('b_BN/beta:0', np.zeros(8, np.float32))])]
    for whole in (False, True):
        p = str(tmp_path / ('w%d.h5' % whole))
        h5io.write_keras_h5(p, layers, whole_model=whole, model_config='{"x": 1}' if whole else None)
        back, attrs = h5io.read_keras_h5(p)
        assert [n for n, _ in back] == [n for n, _ in layers]
        for (_, ws), (_, ws0) in zip(back, layers):
            assert [w for w, _ in ws] == [w for w, _ in ws0]
            assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(ws, ws0))
        assert attrs['keras_version'] == '2.11.0' and (not whole or attrs['model_config'] == '{"x": 1}')
    # more names than one object header holds
    big = [('L%03d_' % i + 'y' * 900, [('v%d:0' % i, np.full(3, i, np.float32))]) for i in range(100)]
    p = str(tmp_path / 'big.h5')
    h5io.write_keras_h5(p, big)
    back, _ = h5io.read_keras_h5(p)
    assert [n for n, _ in back] == [n for n, _ in big] and float(back[99][1][0][1][0]) == 99.0


def test_not_hdf5_is_reported(h5io, tmp_path):
    p = tmp_path / 'junk.h5'
    p.write_bytes(b'not an hdf5 file')
    with pytest.raises(IOError):
        h5io.read_keras_h5(str(p))


@pytest.mark.parametrize('model_type', ['mobilenetv2_lite', 'mobilenetv3large'])
def test_model_h5_round_trip_and_topological_load(h5io, tmp_path, model_type):
    """model.save('*.h5') / save_weights -> a fresh model (other seed) -> load_weights: identical weights; the
    topological loader ignores names and insists on the layer count (Keras' error message)"""
    pkg = load_pkg()
    a = pkg.get_deeplabv3p_model(model_type, 21, (65, 65), 16, training=True, seed=1)
    wa = a.get_weights_by_name()
    for whole in (True, False):
        p = str(tmp_path / ('m%d.h5' % whole))
        (a.save if whole else a.save_weights)(p)
        layers, attrs = h5io.read_keras_h5(p)
        assert [n for n, _ in layers] == [l.name for l in a.layers]                # weightless layers listed too, Keras order
        first = next(ws for _, ws in layers if ws)
        assert first[0][0].endswith('kernel:0')                                      # Keras weight names: '<layer>/kernel:0'
        for by_name in (False, True):
            b = pkg.get_deeplabv3p_model(model_type, 21, (65, 65), 16, training=True, seed=2)
            assert any(not np.array_equal(v, wa[k]) for k, v in b.get_weights_by_name().items())
            b.load_weights(p, by_name=by_name)
            wb = b.get_weights_by_name()
            assert wb.keys() == wa.keys() and all(np.array_equal(wb[k], wa[k]) for k in wa)
    c = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, training=True)
    with pytest.raises(ValueError, match='weight file containing'):
        c.load_weights(p, by_name=False)
    # the factory's weights_path argument goes through the same loader (model.py:102-104)
    d = pkg.get_deeplabv3p_model(model_type, 21, (65, 65), 16, weights_path=p, training=True, seed=3)
    assert all(np.array_equal(v, wa[k]) for k, v in d.get_weights_by_name().items())


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason='no interpreter with h5py in this environment')
def test_h5py_reads_what_h5io_writes(h5io, tmp_path):
    """the independent check: the real h5py sees Keras' structure in a file written by h5io, and Keras' own
    attribute decoding (`n.decode('utf8')` over `group.attrs[name]`) works on it"""
    rng = np.random.default_rng(5)
    layers = [('image_input', []), ('Conv', [('Conv/kernel:0', rng.standard_normal((3, 3, 3, 8)).astype(np.float32))]),
              ('Conv_BN', [('Conv_BN/gamma:0', rng.standard_normal(8).astype(np.float32)),
                           ('Conv_BN/beta:0', rng.standard_normal(8).astype(np.float32))]), ('re_lu', [])]
    p = str(tmp_path / 'ours.h5')
    h5io.write_keras_h5(p, layers, whole_model=True, model_config='{"class_name": "X"}')
    np.savez(str(tmp_path / 'want.npz'), **{wn: a for _, ws in layers for wn, a in ws})
    code = r'''
import sys, h5py, numpy as np
f = h5py.File(sys.argv[1], 'r')
want = np.load(sys.argv[2])
assert f.attrs['backend'] in (b'tensorflow', 'tensorflow') and 'model_config' in f.attrs
g = f['model_weights']
names = [n.decode('utf8') if hasattr(n, 'decode') else n for n in g.attrs['layer_names']]
assert names == ['image_input', 'Conv', 'Conv_BN', 're_lu'], names
seen = 0
for n in names:
    wn = [x.decode('utf8') if hasattr(x, 'decode') else x for x in g[n].attrs['weight_names']]
    for w in wn:
        d = g[n][w]
        assert d.dtype == np.float32 and np.array_equal(np.asarray(d), want[w]), w
        seen += 1
assert seen == 3 and len(g['re_lu'].attrs['weight_names']) == 0
print('H5PY_OK')
'''
    r = subprocess.run([H5PY_PYTHON, '-c', code, p, str(tmp_path / 'want.npz')], capture_output=True, text=True)
    assert 'H5PY_OK' in r.stdout, r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------
# Keras layer order of the branched graphs: `model.layers`, and with it the order of the weighted
# layers in a `.h5` file and the pairing of `load_weights(by_name=False)`, is by DECREASING depth from the output with
# the depth-first visiting order as tie-break (keras/engine/functional.py `_map_graph_network`), not creation order.
def _weighted_names(model):
    return [l.name for l in model.layers if l.params]


def test_keras_layer_order_mobilenetv2_aspp_and_decoder():
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16)
    names = _weighted_names(m)
    i = names.index('aspp1_depthwise')
    assert names[i - 1] == 'expanded_conv_16_project_BN'
    # hand-derived from reference layers.py:131-161 (depths counted back from the Concatenate):
    #   +6 the three depthwise convs, +5 their BNs (and the pooling layer), +4 image_pooling, +3 image_pooling_BN and the
    #   four 1x1 convs, +2 their BNs; branches in the order of Concatenate([b4, b0, b1, b2, b3])
    assert names[i:i + 18] == [
        'aspp1_depthwise', 'aspp2_depthwise', 'aspp3_depthwise',
        'aspp1_depthwise_BN', 'aspp2_depthwise_BN', 'aspp3_depthwise_BN',
        'image_pooling',
        'image_pooling_BN', 'aspp0', 'aspp1_pointwise', 'aspp2_pointwise', 'aspp3_pointwise',
        'aspp0_BN', 'aspp1_pointwise_BN', 'aspp2_pointwise_BN', 'aspp3_pointwise_BN',
        'concat_projection', 'concat_projection_BN']
    # the decoder's skip projection hangs off expanded_conv_2 but sits 3 layers above the decoder Concatenate
    assert names[i + 18:i + 20] == ['feature_projection0', 'feature_projection0_BN']
    assert names[i + 20] == 'decoder_conv0_depthwise' and names[-1] == 'conv_upsample'
    assert m.layers[0].name == 'image_input' and m.layers[-1].name == 'pred_mask'
    # the first backbone_len layers are exactly the backbone (what freeze_level=1 freezes, model.py:106-110)
    head = {'image_pooling', 'aspp0', 'aspp1_depthwise', 'feature_projection0', 'decoder_conv0_depthwise', 'conv_upsample'}
    assert not head & {l.name for l in m.layers[:m.backbone_len]}
    assert sum(1 for l in m.layers[:m.backbone_len] if l.name.startswith(('expanded_conv', 'Conv'))) == \
        sum(1 for l in m.layers if l.name.startswith(('expanded_conv', 'Conv')))


def test_keras_layer_order_xception_shortcut_interleaves():
    pkg = load_pkg()
    m = pkg.get_deeplabv3p_model('xception', 21, (65, 65), 16)
    names = _weighted_names(m)
    i = names.index('entry_flow_block1_separable_conv3_depthwise')
    # add([residual, shortcut]) (deeplabv3p_xception.py:85): the shortcut conv ties with separable_conv3_pointwise at
    # depth +2 and its BN with separable_conv3_pointwise_BN at +1; the residual branch is visited first
    assert names[i:i + 6] == ['entry_flow_block1_separable_conv3_depthwise', 'entry_flow_block1_separable_conv3_depthwise_BN',
                              'entry_flow_block1_separable_conv3_pointwise', 'entry_flow_block1_shortcut',
                              'entry_flow_block1_separable_conv3_pointwise_BN', 'entry_flow_block1_shortcut_BN']


def test_h5_topological_round_trip_of_a_branched_model(tmp_path):
    """save_weights -> load_weights(by_name=False) of full MobileNetV2 / by-name into a fresh model: same weights; the
    file lists the layers in Keras order"""
    pkg = load_pkg()
    h5io = load_pkg('h5io')
    try:
        h5io.lib()
    except ImportError:
        pytest.skip('libhdf5 not available')
    a = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, seed=1)
    b = pkg.get_deeplabv3p_model('mobilenetv2', 21, (65, 65), 16, seed=2)
    path = str(tmp_path / 'w.h5')
    a.save_weights(path)
    file_layers, _ = h5io.read_keras_h5(path)
    assert [n for n, ws in file_layers if ws] == _weighted_names(a)
    b.load_weights(path, by_name=False)
    wa, wb = a.get_weights_by_name(), b.get_weights_by_name()
    assert all(np.array_equal(wa[k], wb[k]) for k in wa)
    assert [w.shape for w in a.get_weights()] == [p.shape for l in a.layers for p in l.params]
