import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DL3P_PW_SMALL_MIN_ROWS', '64')
from conftest import load_pkg
import torch
from test_model_gpu import _pair, _data
mt, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N, C = 2, 21
x, y = _data(N, H, W, C, seed=11)
def grads(env):
    for k in ('DL3P_FOLD_APPLY', 'DL3P_GRAD_ALIAS', 'DL3P_FUSE_BN_BWD', 'DL3P_BATCHED_WGRAD'):
        os.environ.pop(k, None)
    os.environ.update(env)
    m, o = _pair(mt, H, W, C)
    m.use_graphs = False
    loss = m.train_on_batch(x, y)
    st = m._store
    return loss, {p.name: np.array(st.get(p, st.G), dtype=np.float64) for p in m.graph.all_params() if p.trainable}, m, o
l1, g1, m, o = grads({})
# oracle
total, ce, logits = o.loss_and_grads(x, y, None) if False else (None, None, None)
for env in ({'DL3P_FOLD_APPLY': '0'}, {'DL3P_GRAD_ALIAS': '0'}, {'DL3P_FUSE_BN_BWD': '0'}, {'DL3P_FUSE_BN_BWD': '1single'}, {'DL3P_BATCHED_WGRAD': '0'},
            {'DL3P_FOLD_APPLY': '0', 'DL3P_GRAD_ALIAS': '0', 'DL3P_FUSE_BN_BWD': '0', 'DL3P_BATCHED_WGRAD': '0'}):
    l0, g0, _, _ = grads(env)
    gmax = max(float(np.abs(a).max()) for a in g0.values())
    rs = sorted(((float(np.abs(g0[n] - g1[n]).max() / (np.abs(g0[n]).max() + 1e-4 * gmax)), n) for n in g0), reverse=True)
    print(env, 'loss', l1, l0, 'worst', rs[:6])
