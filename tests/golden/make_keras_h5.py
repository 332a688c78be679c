"""Writes the Keras-HDF5 golden files with the REAL h5py (run it with an interpreter that has h5py, here
/opt/conda/bin/python3.9; h5py 3.3.0 / HDF5 1.10.6 made the committed files):

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py tests/golden

The h5py calls restate Keras 2.11 `saving/legacy/hdf5_format.py`:
  save_weights_to_hdf5_group  -> attrs layer_names / backend / keras_version, one group per layer with attrs
                                 weight_names and one dataset per weight (`g.create_dataset(name, shape, dtype)`)
  save_attributes_to_hdf5_group -> numpy 'S' arrays, split into <name>%d chunks above 64512 bytes
  save_model_to_hdf5          -> the same tree under `model_weights`, `model_config` JSON as a (variable-length) str
Weights are a deterministic function of their name so the test can regenerate the expected values."""
import json
import sys
import zlib

import h5py
import numpy as np

HDF5_OBJECT_HEADER_LIMIT = 64512


def weight(name, shape):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return rng.standard_normal(shape).astype(np.float32)


def save_attributes_to_hdf5_group(group, name, data):
    bad = [x for x in data if len(x) > HDF5_OBJECT_HEADER_LIMIT]
    assert not bad
    data_npy = np.asarray(data)
    num_chunks = 1
    chunked = np.array_split(data_npy, num_chunks)
    while any(x.nbytes > HDF5_OBJECT_HEADER_LIMIT for x in chunked):
        num_chunks += 1
        chunked = np.array_split(data_npy, num_chunks)
    if num_chunks > 1:
        for i, c in enumerate(chunked):
            group.attrs['%s%d' % (name, i)] = c
    else:
        group.attrs[name] = data


def save_weights_to_hdf5_group(f, layers):
    save_attributes_to_hdf5_group(f, 'layer_names', [n.encode('utf8') for n, _ in layers])
    f.attrs['backend'] = 'tensorflow'.encode('utf8')
    f.attrs['keras_version'] = '2.11.0'.encode('utf8')
    for lname, ws in layers:
        g = f.create_group(lname)
        names = [wn.encode('utf8') for wn, _ in ws]
        save_attributes_to_hdf5_group(g, 'weight_names', names)
        for wn, shape in ws:
            val = weight(wn, shape)
            d = g.create_dataset(wn, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


# a miniature of the reference's layer list: weightless layers in between, Conv2D with and without bias,
# DepthwiseConv2D, BatchNormalization (gamma, beta, moving_mean, moving_variance), the 1x1 head with bias
SMALL = [
    ('image_input', []),
    ('Conv', [('Conv/kernel:0', (3, 3, 3, 8))]),
    ('Conv_BN', [('Conv_BN/gamma:0', (8,)), ('Conv_BN/beta:0', (8,)), ('Conv_BN/moving_mean:0', (8,)),
                 ('Conv_BN/moving_variance:0', (8,))]),
    ('re_lu', []),
    ('expanded_conv_depthwise', [('expanded_conv_depthwise/depthwise_kernel:0', (3, 3, 8, 1))]),
    ('expanded_conv_depthwise_BN', [('expanded_conv_depthwise_BN/gamma:0', (8,)), ('expanded_conv_depthwise_BN/beta:0', (8,)),
                                    ('expanded_conv_depthwise_BN/moving_mean:0', (8,)),
                                    ('expanded_conv_depthwise_BN/moving_variance:0', (8,))]),
    ('expanded_conv_project', [('expanded_conv_project/kernel:0', (1, 1, 8, 4))]),
    ('dropout', []),
    ('conv_upsample', [('conv_upsample/kernel:0', (1, 1, 4, 21)), ('conv_upsample/bias:0', (21,))]),
    ('pred_resize', []),
    ('pred_mask', []),
]
# enough long names to push layer_names over the object-header limit (chunked attributes)
MANY = [('unit_%03d_' % i + 'x' * 990, [('w%d/gamma:0' % i, (2,))]) for i in range(70)]


def main(out):
    with h5py.File(out + '/keras_weights_small.h5', 'w') as f:
        save_weights_to_hdf5_group(f, SMALL)
    with h5py.File(out + '/keras_model_small.h5', 'w') as f:                 # model.save(): save_model_to_hdf5
        f.attrs['keras_version'] = '2.11.0'
        f.attrs['backend'] = 'tensorflow'
        f.attrs['model_config'] = json.dumps({'class_name': 'Functional', 'config': {'name': 'model'}})
        save_weights_to_hdf5_group(f.create_group('model_weights'), SMALL)
    with h5py.File(out + '/keras_weights_chunked_names.h5', 'w') as f:
        save_weights_to_hdf5_group(f, MANY)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '.')
