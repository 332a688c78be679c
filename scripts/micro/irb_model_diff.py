"""per-parameter gradient difference of one train step with / without the fused inverted-residual blocks (debug aid)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import test_model_gpu as TM  # noqa: E402

mt, H, W = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('mobilenetv2', 65, 65)
N, C = 2, 21
x, y = TM._data(N, H, W, C, seed=17)


def run(env):
    for k in ('DL3P_IRB', 'DL3P_IRB_MIN_ROWS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    torch.manual_seed(0)
    m, _ = TM._pair(mt, H, W, C)
    m.use_graphs = False
    loss = m.train_on_batch(x, y)
    st = m._store
    return loss, [(p.name, np.array(st.get(p, st.G), dtype=np.float64)) for p in m.graph.all_params() if p.trainable]


l1, g1 = run({'DL3P_IRB_MIN_ROWS': '1'})
l0, g0 = run({'DL3P_IRB': '0'})
print('loss', l1, l0)
for (n, a), (_, b) in zip(g0, g1):
    r = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-12))
    if r > 1e-4:
        print('%-50s %.3e  |ref| %.3e' % (n, r, np.abs(a).max()))
