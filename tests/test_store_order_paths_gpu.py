"""Round-6 kernels that only reorder memory operations must not change a bit: the depthwise forward's counted-wait rows
(DL3P_DW_FAST_ROWS: csrc/dwconv.hip dw_fwd_seg FAST), the segment form of the bilinear upsampling (DL3P_RESIZE_STRIP: csrc/resize_head.hip
resize_fwd_seg_kernel) and the eight-column window of its backward (DL3P_RESIZE_TIGHT) against the paths they replace
(reference call sites: /root/reference deeplabv3p/models/layers.py:100-101 DepthwiseConv2D, :207 decoder_resize).

The switches are read once per process, so each side runs in its own interpreter and leaves SHA-256 digests of its outputs."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, importlib, json, sys
import torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + '/tests')
from conftest import load_pkg
ops = importlib.import_module(load_pkg().__name__ + '.ops')
dev = 'cuda:0'
out = {}
def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
torch.manual_seed(7)
# depthwise forward with the producer's BatchNorm + ReLU6 prologue and the statistics epilogue: strides, rates, strip remainders
for (N, H, W, C, s, r) in [(2, 33, 33, 320, 1, 1), (1, 65, 65, 192, 1, 1), (2, 33, 33, 960, 1, 2), (1, 129, 129, 144, 2, 1),
                           (1, 40, 37, 24, 1, 1), (1, 33, 33, 320, 1, 6), (2, 17, 19, 304, 1, 1)]:
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(3, 3, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    part = ops.new_partials(C, dev)
    y, rows = ops.dwconv2d_fwd(x, w, s, r, 'same', sc, sh, ops.ACT_RELU6, partials=part)
    out['dw %%s' %% ((N, H, W, C, s, r),)] = [digest(y), digest(part[:rows * 2 * C])]
for (N, h, w, C, H, W) in [(2, 33, 33, 256, 129, 129), (1, 9, 13, 256, 33, 50), (1, 97, 97, 256, 193, 193), (3, 5, 7, 512, 40, 29),
                           (1, 1, 1, 256, 33, 33)]:
    x = torch.randn(N, h, w, C, device=dev)
    buf = torch.zeros(N, H, W, C + 48, device=dev)
    ops.resize_bilinear_fwd(x, H, W, out=buf[..., :C])
    g = torch.randn(N, H, W, C, device=dev)
    gx = ops.resize_bilinear_bwd(g, h, w)
    out['resize %%s' %% ((N, h, w, C, H, W),)] = [digest(buf), digest(gx)]
print('DIGESTS ' + json.dumps(out))
'''


def _run(env):
    import torch
    if torch.cuda.is_available():
        torch.cuda.empty_cache()          # (the children run on the same GPU)
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('DIGESTS ')][-1]
    return json.loads(line[len('DIGESTS '):])


def test_reordered_memory_paths_leave_every_bit_where_it_was():
    new = _run({'DL3P_DW_FAST_ROWS': '2', 'DL3P_RESIZE_STRIP': '1', 'DL3P_RESIZE_TIGHT': '1'})
    old = _run({'DL3P_DW_FAST_ROWS': '0', 'DL3P_RESIZE_STRIP': '0', 'DL3P_RESIZE_TIGHT': '0'})
    assert new.keys() == old.keys() and len(new) == 12
    diff = [k for k in new if new[k] != old[k]]
    assert not diff, diff
