"""Keras HDF5 weight files (`model.save_weights('x.h5')` / `model.save('x.h5')`, the format of the reference's
released checkpoints: README.md:58-61, train.py:247, model.py:102-104) read and written through the HDF5 C library
with ctypes -- h5py is not part of this image's interpreter, libhdf5 is (DL3P_HDF5_LIB overrides the search).

Layout restated from Keras 2.11 `saving/legacy/hdf5_format.py` (`save_weights_to_hdf5_group` /
`load_weights_from_hdf5_group[_by_name]`):

    /                       attrs: layer_names = [b'image_input', b'Conv', ...]   (every layer, also weightless ones)
                                   backend = b'tensorflow', keras_version = b'2.11.0'
    /<layer>                attrs: weight_names = [b'Conv/kernel:0', ...]         (trainable, then non-trainable)
    /<layer>/<weight_name>  float32 dataset in the Keras shape (HWIO kernels, (3,3,C,1) depthwise kernels,
                            BatchNormalization as gamma, beta, moving_mean, moving_variance)

A whole-model file (`model.save`) keeps the same tree under the group `model_weights`; string attributes too large
for one object header are split into `<name>0`, `<name>1`, ... chunks (`save_attributes_to_hdf5_group`).
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_hsize = ctypes.c_uint64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5S_SCALAR = 0
_H5T_STRING, _H5T_FLOAT, _H5T_INTEGER = 3, 1, 0
_H5T_VARIABLE = ctypes.c_size_t(-1).value
_HDR_LIMIT = 64512          # Keras' HDF5_OBJECT_HEADER_LIMIT


class H5Error(IOError):
    pass


_lib = None


def _find_lib():
    cands = [os.environ.get('DL3P_HDF5_LIB'), ctypes.util.find_library('hdf5'), '/opt/conda/lib/libhdf5.so',
             '/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so', '/usr/lib/x86_64-linux-gnu/libhdf5_serial.so']
    for c in cands:
        if not c:
            continue
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    raise ImportError('libhdf5 not found (set DL3P_HDF5_LIB); Keras .h5 files need the HDF5 C library, '
                      '.npz checkpoints do not')


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = _find_lib()
    sig = {
        'H5open': (ctypes.c_int, []),
        'H5Eset_auto2': (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
        'H5Fopen': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
        'H5Fcreate': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
        'H5Fclose': (ctypes.c_int, [_hid]),
        'H5Gopen2': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Gcreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid]),
        'H5Gclose': (ctypes.c_int, [_hid]),
        'H5Lexists': (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
        'H5Aexists': (ctypes.c_int, [_hid, ctypes.c_char_p]),
        'H5Aopen': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Acreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid]),
        'H5Aread': (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
        'H5Awrite': (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
        'H5Aclose': (ctypes.c_int, [_hid]),
        'H5Aget_type': (_hid, [_hid]),
        'H5Aget_space': (_hid, [_hid]),
        'H5Dopen2': (_hid, [_hid, ctypes.c_char_p, _hid]),
        'H5Dcreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        'H5Dread': (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        'H5Dwrite': (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        'H5Dclose': (ctypes.c_int, [_hid]),
        'H5Dget_space': (_hid, [_hid]),
        'H5Dget_type': (_hid, [_hid]),
        'H5Dvlen_reclaim': (ctypes.c_int, [_hid, _hid, _hid, ctypes.c_void_p]),
        'H5Sget_simple_extent_ndims': (ctypes.c_int, [_hid]),
        'H5Sget_simple_extent_dims': (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        'H5Sget_simple_extent_npoints': (ctypes.c_int64, [_hid]),
        'H5Screate_simple': (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        'H5Screate': (_hid, [ctypes.c_int]),
        'H5Sclose': (ctypes.c_int, [_hid]),
        'H5Tcopy': (_hid, [_hid]),
        'H5Tset_size': (ctypes.c_int, [_hid, ctypes.c_size_t]),
        'H5Tget_size': (ctypes.c_size_t, [_hid]),
        'H5Tget_class': (ctypes.c_int, [_hid]),
        'H5Tis_variable_str': (ctypes.c_int, [_hid]),
        'H5Tclose': (ctypes.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if L.H5open() < 0:
        raise H5Error('H5open failed')
    L.H5Eset_auto2(0, None, None)          # errors are reported through return codes, not printed stacks
    L.t_f32 = _hid.in_dll(L, 'H5T_IEEE_F32LE_g').value
    L.t_native_f32 = _hid.in_dll(L, 'H5T_NATIVE_FLOAT_g').value
    L.t_native_f64 = _hid.in_dll(L, 'H5T_NATIVE_DOUBLE_g').value
    L.t_native_i64 = _hid.in_dll(L, 'H5T_NATIVE_LLONG_g').value
    L.t_c_s1 = _hid.in_dll(L, 'H5T_C_S1_g').value
    _lib = L
    return L


def _ok(v, what):
    if v < 0:
        raise H5Error('HDF5 call failed: ' + what)
    return v


def _dims(L, space):
    nd = _ok(L.H5Sget_simple_extent_ndims(space), 'ndims')
    d = (_hsize * max(nd, 1))()
    if nd:
        _ok(L.H5Sget_simple_extent_dims(space, d, None), 'dims')
    return tuple(int(d[i]) for i in range(nd))


# ------------------------------------------------------------------------------------------------ reading
def _read_attr(L, loc, name):
    """-> list of bytes for string attributes, numpy array for numeric ones, None when absent"""
    if L.H5Aexists(loc, name.encode()) <= 0:
        return None
    a = _ok(L.H5Aopen(loc, name.encode(), _H5P_DEFAULT), 'H5Aopen ' + name)
    t = L.H5Aget_type(a)
    s = L.H5Aget_space(a)
    try:
        shape = _dims(L, s)
        n = int(L.H5Sget_simple_extent_npoints(s))
        cls = L.H5Tget_class(t)
        if cls == _H5T_STRING:
            if n == 0:
                return []
            if L.H5Tis_variable_str(t) > 0:
                # memory type = the file type: a variable-length string keeps its character set (h5py writes
                # ASCII for bytes, UTF-8 for str; HDF5 refuses to convert between the two)
                ptrs = (ctypes.c_char_p * n)()
                _ok(L.H5Aread(a, t, ptrs), 'H5Aread ' + name)
                out = [bytes(p) if p is not None else b'' for p in ptrs]
                L.H5Dvlen_reclaim(t, s, _H5P_DEFAULT, ptrs)
            else:
                size = int(L.H5Tget_size(t))
                buf = ctypes.create_string_buffer(size * n)
                _ok(L.H5Aread(a, t, buf), 'H5Aread ' + name)
                raw = buf.raw
                out = [raw[i * size:(i + 1) * size].split(b'\0', 1)[0] for i in range(n)]
            return out if shape else out[0:1]
        if n == 0:
            return np.zeros(shape, np.float64)
        if cls == _H5T_INTEGER:
            arr = np.empty(shape, np.int64)
            _ok(L.H5Aread(a, L.t_native_i64, arr.ctypes.data_as(ctypes.c_void_p)), 'H5Aread ' + name)
        else:
            arr = np.empty(shape, np.float64)
            _ok(L.H5Aread(a, L.t_native_f64, arr.ctypes.data_as(ctypes.c_void_p)), 'H5Aread ' + name)
        return arr
    finally:
        L.H5Sclose(s)
        L.H5Tclose(t)
        L.H5Aclose(a)


def _read_names(L, loc, name):
    """Keras `load_attributes_from_hdf5_group`: the attribute, or its `<name>0`, `<name>1`, ... chunks"""
    v = _read_attr(L, loc, name)
    if v is not None:
        return [x.decode('utf8') for x in v] if isinstance(v, list) else []
    out, i = [], 0
    while True:
        v = _read_attr(L, loc, '%s%d' % (name, i))
        if v is None:
            break
        out += [x.decode('utf8') for x in v]
        i += 1
    if i == 0:
        raise H5Error('attribute %r not found: not a Keras weight file?' % name)
    return out


def _read_dataset(L, loc, path):
    d = _ok(L.H5Dopen2(loc, path.encode(), _H5P_DEFAULT), 'H5Dopen ' + path)
    s = L.H5Dget_space(d)
    try:
        arr = np.empty(_dims(L, s), np.float32)
        if arr.size:
            _ok(L.H5Dread(d, L.t_native_f32, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)),
                'H5Dread ' + path)
        return arr
    finally:
        L.H5Sclose(s)
        L.H5Dclose(d)


def read_keras_h5(path):
    """-> (layers, attrs): layers = [(layer_name, [(weight_name, float32 array), ...]), ...] in file order (every layer of
    `layer_names`, weightless ones with an empty list); attrs = {'backend': ..., 'keras_version': ..., 'model_config': ...}
    as far as present.  Accepts weight-only files and whole-model files (tree under `model_weights`)."""
    L = lib()
    f = L.H5Fopen(path.encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
    if f < 0:
        raise H5Error('cannot open %s as HDF5' % path)
    root = None
    try:
        attrs = {}
        for k in ('backend', 'keras_version', 'model_config', 'training_config'):
            v = _read_attr(L, f, k)
            if isinstance(v, list) and v:
                attrs[k] = v[0].decode('utf8')
        top = f
        if L.H5Aexists(f, b'layer_names') <= 0 and L.H5Aexists(f, b'layer_names0') <= 0 \
                and L.H5Lexists(f, b'model_weights', _H5P_DEFAULT) > 0:
            root = _ok(L.H5Gopen2(f, b'model_weights', _H5P_DEFAULT), 'open model_weights')
            top = root
            for k in ('backend', 'keras_version'):
                v = _read_attr(L, top, k)
                if isinstance(v, list) and v:
                    attrs.setdefault(k, v[0].decode('utf8'))
        layers = []
        for lname in _read_names(L, top, 'layer_names'):
            g = _ok(L.H5Gopen2(top, lname.encode(), _H5P_DEFAULT), 'open group ' + lname)
            try:
                ws = [(wn, _read_dataset(L, g, wn)) for wn in _read_names(L, g, 'weight_names')]
            finally:
                L.H5Gclose(g)
            layers.append((lname, ws))
        return layers, attrs
    finally:
        if root is not None:
            L.H5Gclose(root)
        L.H5Fclose(f)


# ------------------------------------------------------------------------------------------------ writing
def _write_str_attr(L, loc, name, values, scalar=False):
    """fixed-length, null-padded byte strings -- what h5py makes of a numpy 'S' array (or of one bytes object)"""
    vals = [v.encode('utf8') if isinstance(v, str) else bytes(v) for v in values]
    if not vals and not scalar:
        # h5py stores `attrs[name] = []` as a float64 attribute of shape (0,)
        dims = (_hsize * 1)(0)
        s = _ok(L.H5Screate_simple(1, dims, None), 'space')
        a = _ok(L.H5Acreate2(loc, name.encode(), L.t_native_f64, s, _H5P_DEFAULT, _H5P_DEFAULT), 'H5Acreate ' + name)
        L.H5Aclose(a)
        L.H5Sclose(s)
        return
    size = max(1, max(len(v) for v in vals))
    t = L.H5Tcopy(L.t_c_s1)
    _ok(L.H5Tset_size(t, size), 'H5Tset_size')
    if scalar:
        s = _ok(L.H5Screate(_H5S_SCALAR), 'space')
    else:
        dims = (_hsize * 1)(len(vals))
        s = _ok(L.H5Screate_simple(1, dims, None), 'space')
    buf = b''.join(v.ljust(size, b'\0') for v in vals)
    a = _ok(L.H5Acreate2(loc, name.encode(), t, s, _H5P_DEFAULT, _H5P_DEFAULT), 'H5Acreate ' + name)
    _ok(L.H5Awrite(a, t, ctypes.c_char_p(buf)), 'H5Awrite ' + name)
    L.H5Aclose(a)
    L.H5Sclose(s)
    L.H5Tclose(t)


def _write_names(L, loc, name, values):
    """Keras `save_attributes_to_hdf5_group`: split into `<name>%d` chunks when the array exceeds the header limit"""
    vals = [v.encode('utf8') for v in values]
    size = max([1] + [len(v) for v in vals])
    if size * len(vals) <= _HDR_LIMIT:
        _write_str_attr(L, loc, name, vals)
        return
    if size > _HDR_LIMIT:
        raise H5Error('a single name exceeds the HDF5 object header limit')
    chunks = 1
    while True:
        parts = np.array_split(np.arange(len(vals)), chunks)
        if all(size * len(p) <= _HDR_LIMIT for p in parts):
            break
        chunks += 1
    for i, p in enumerate(parts):
        _write_str_attr(L, loc, '%s%d' % (name, i), [vals[j] for j in p])


def _open_groups(L, loc, parts):
    """open (or create) the chain of groups `parts` below loc, as h5py does for names with '/' in them (MobileNetV3's
    Keras layer names: 'expanded_conv/depthwise/Conv'); -> handles, innermost last"""
    opened, cur = [], loc
    try:
        for pth in parts:
            if L.H5Lexists(cur, pth.encode(), _H5P_DEFAULT) > 0:
                g = _ok(L.H5Gopen2(cur, pth.encode(), _H5P_DEFAULT), 'open ' + pth)
            else:
                g = _ok(L.H5Gcreate2(cur, pth.encode(), _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT), 'create ' + pth)
            opened.append(g)
            cur = g
    except H5Error:
        for g in reversed(opened):
            L.H5Gclose(g)
        raise
    return opened


def _write_dataset(L, loc, path, arr):
    arr = np.ascontiguousarray(arr, dtype='<f4')
    parts = path.split('/')
    opened = _open_groups(L, loc, parts[:-1])
    cur = opened[-1] if opened else loc
    try:
        if arr.ndim:
            dims = (_hsize * arr.ndim)(*arr.shape)
            s = _ok(L.H5Screate_simple(arr.ndim, dims, None), 'space')
        else:
            s = _ok(L.H5Screate(_H5S_SCALAR), 'space')
        d = _ok(L.H5Dcreate2(cur, parts[-1].encode(), L.t_f32, s, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT),
                'H5Dcreate ' + path)
        if arr.size:
            _ok(L.H5Dwrite(d, L.t_native_f32, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)),
                'H5Dwrite ' + path)
        L.H5Dclose(d)
        L.H5Sclose(s)
    finally:
        for g in reversed(opened):
            L.H5Gclose(g)


def write_keras_h5(path, layers, whole_model=False, model_config=None, keras_version='2.11.0'):
    """layers = [(layer_name, [(weight_name, array), ...]), ...] in topological order, weightless layers included with an
    empty list (Keras lists them too).  whole_model=True puts the tree under `model_weights` as `model.save` does and
    stores `model_config` (a JSON string) next to it."""
    L = lib()
    f = L.H5Fcreate(path.encode(), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
    if f < 0:
        raise H5Error('cannot create %s' % path)
    top = f
    try:
        if whole_model:
            _write_str_attr(L, f, 'keras_version', [keras_version], scalar=True)
            _write_str_attr(L, f, 'backend', ['tensorflow'], scalar=True)
            if model_config is not None:
                _write_str_attr(L, f, 'model_config', [model_config], scalar=True)
            top = _ok(L.H5Gcreate2(f, b'model_weights', _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT), 'model_weights')
        _write_names(L, top, 'layer_names', [n for n, _ in layers])
        _write_str_attr(L, top, 'backend', ['tensorflow'], scalar=True)
        _write_str_attr(L, top, 'keras_version', [keras_version], scalar=True)
        for lname, ws in layers:
            chain = _open_groups(L, top, lname.split('/'))
            g = chain[-1]
            try:
                _write_names(L, g, 'weight_names', [wn for wn, _ in ws])
                for wn, arr in ws:
                    _write_dataset(L, g, wn, arr)
            finally:
                for h in reversed(chain):
                    L.H5Gclose(h)
    finally:
        if top != f:
            L.H5Gclose(top)
        L.H5Fclose(f)
