"""compare dL/d(raw conv output) for every conv between the HIP executor and the oracle"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import np_net
from test_model_gpu import _pair, _data, _rel
rec = {}
_c, _d = np_net.Net.conv2d, np_net.Net.dwconv2d
def conv2d(self, x, filters, k, name, *a, **kw):
    y = _c(self, x, filters, k, name, *a, **kw); rec[name] = (x, y); return y
def dwconv2d(self, x, k, name, *a, **kw):
    y = _d(self, x, k, name, *a, **kw); rec[name] = (x, y); return y
np_net.Net.conv2d, np_net.Net.dwconv2d = conv2d, dwconv2d
mt = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 65
N, C = 2, 21
m, o = _pair(mt, H, W, C)
m.use_graphs = False
x, y = _data(N, H, W, C, seed=3)
loss = m.train_on_batch(x, y)
ex = m._executor(N, True)
drop = [op for op in m.graph.ops if op.kind == 'materialize' and op.rate > 0][0]
mask = ex.dropout_mask(drop).cpu().numpy()
rec.clear()
total, ce, logits = o.loss_and_grads(x, y, {'aspp_dropout': mask})
print('loss', loss, ce)
for op in reversed(m.graph.ops):
    if op.kind not in ('conv_pw', 'conv_dw', 'conv_dense'): continue
    xv, yv = rec[op.name]
    z = ex.view(op.out).cpu().numpy()[..., :yv.v.shape[-1]]
    dz = ex.view(op.out, grad=True).cpu().numpy()[..., :yv.v.shape[-1]]
    print('%-40s z rel %.1e   dz rel %.1e  |dz| %.1e' % (op.name, _rel(z, yv.v), _rel(dz, yv.g), np.abs(yv.g).max()))
for name in ['decoder_conv1_pointwise', 'decoder_conv1_depthwise']:
    op = [o_ for o_ in m.graph.ops if getattr(o_, 'name', '') == name and o_.kind.startswith('conv')][0]
    xv, yv = rec[name]
    dz = ex.view(op.out, grad=True).cpu().numpy().astype(np.float64)
    err = np.abs(dz - yv.g)
    mx = np.abs(yv.g).max()
    print(name, 'frac elems err>1e-3*max:', (err > 1e-3 * mx).mean(), 'per-channel max err (top5):',
          np.sort(err.reshape(-1, err.shape[-1]).max(0))[-5:] / mx, 'median ch err', np.median(err.reshape(-1, err.shape[-1]).max(0)) / mx)
    idx = np.unravel_index(np.argmax(err), err.shape)
    print('  worst at', idx, 'gpu', dz[idx], 'ref', yv.g[idx])
