"""ctypes binding of libdl3p.so.  Signatures are parsed from include/dl3p.h (the single source of
truth for the C ABI) so the Python side cannot drift from the header.

The product path has NO fallback: if the HIP library is missing, importing ops fails loudly.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, '..', 'include', 'dl3p.h')
LIBPATH = os.path.join(HERE, 'libdl3p.so')

_CT = {
    'const float*': ctypes.c_void_p, 'float*': ctypes.c_void_p,
    'const double*': ctypes.c_void_p, 'double*': ctypes.c_void_p,
    'const int64_t*': ctypes.c_void_p, 'int64_t*': ctypes.c_void_p,
    'void*': ctypes.c_void_p, 'const void*': ctypes.c_void_p, 'void**': ctypes.POINTER(ctypes.c_void_p), 'int*': ctypes.POINTER(ctypes.c_int), 'unsigned long long*': ctypes.c_void_p, 'int32_t*': ctypes.c_void_p, 'uint32_t*': ctypes.c_void_p, 'const unsigned char*': ctypes.c_void_p, 'unsigned char*': ctypes.c_void_p, 'uint8_t*': ctypes.c_void_p, 'const uint8_t*': ctypes.c_void_p, 'const int*': ctypes.c_void_p, 'float*': ctypes.c_void_p,
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'size_t': ctypes.c_size_t, 'uint64_t': ctypes.c_uint64,
    'const char*': ctypes.c_char_p, 'void': None,
}


def parse_header(path=HEADER):
    """-> {name: (restype_str, [argtype_str, ...])} for every dl3p_* prototype in the header"""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)
    src = re.sub(r'\benum\s*\{[^}]*\}\s*;', '', src)
    src = src.replace('extern "C" {', '')
    protos = {}
    for m in re.finditer(r'([\w \*]+?)\b(dl3p_\w+)\s*\(([^)]*)\)\s*;', src):
        ret = ' '.join(m.group(1).split())
        name = m.group(2)
        args = []
        for a in m.group(3).split(','):
            a = ' '.join(a.split())
            if a in ('void', ''):
                continue
            mm = re.match(r'(.*?)(\w+)$', a)
            t = mm.group(1).strip().replace(' *', '*')
            args.append(t)
        protos[name] = (ret, args)
    return protos


class Dl3pError(RuntimeError):
    pass


class Lib:
    def __init__(self, path=LIBPATH):
        if not os.path.exists(path):
            raise ImportError(
                'libdl3p.so not found at %s -- build it with `python __graft_entry__.py` '
                '(there is no CPU fallback for the HIP path)' % path)
        # torch ships its own HIP / HSA runtime; it has to be the one already in the process when libdl3p.so is
        # loaded (the loader then binds libdl3p's libamdhip64 dependency to it).  Loaded the other way round the
        # process ends up with two HSA runtimes and the second one sees no device.
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)
            fn.restype = _CT[ret]
            fn.argtypes = [_CT[a] for a in args]
            setattr(self, name[len('dl3p_'):], self._wrap(name, fn, ret))

    def _wrap(self, name, fn, ret):
        if ret != 'int' or name in ('dl3p_version', 'dl3p_device_cus', 'dl3p_head_train_supported', 'dl3p_head_train_rows_supported', 'dl3p_stem_conv_supported', 'dl3p_conv2d_gemm_supported', 'dl3p_reduce_rows_variant', 'dl3p_reduce_rows_block_elements', 'dl3p_pwconv_bwd_weight_bn_supported', 'dl3p_dwconv2d_bwd_weight_bn_supported', 'dl3p_pwconv_sb_supported', 'dl3p_pwconv_sb_pays', 'dl3p_pwconv_bwd_data_sb_apply_supported', 'dl3p_conv2d_gemm_sb_supported', 'dl3p_conv2d_gemm_sb_pays', 'dl3p_irb_supported', 'dl3p_pwconv_fwd_splitk_plan', 'dl3p_irb_bwd_supported', 'dl3p_get_option', 'dl3p_irb_get_plan', 'dl3p_irb_cov_rows_max', 'dl3p_dw_upsampled_input_supported'):
            return fn
        err = self.cdll.dl3p_last_error_string
        err.restype = ctypes.c_char_p

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise Dl3pError('%s failed (%d): %s' % (name, rc, err().decode()))
            return rc
        call.raw = fn
        call.__name__ = name
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        # DL3P_LIB_OVERRIDE: an A/B build of the same sources (scripts/micro/build_variant.sh), for running the
        # parity tests against it; never a different implementation
        _lib = Lib(os.environ['DL3P_LIB_OVERRIDE']) if os.environ.get('DL3P_LIB_OVERRIDE') else Lib()
    return _lib
