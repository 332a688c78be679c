import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from test_model_gpu import _pair, _data, _rel, _act_derivs
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
o.net.act_derivs = _act_derivs(m, ex)
total, ce, logits = o.loss_and_grads(x, y, {'aspp_dropout': mask})
print('loss', loss, ce)
st = m._store
for p in reversed(m.graph.all_params()):
    if not p.trainable: continue
    g = st.get(p, st.G); gref = o.net.grads[p.name]
    r = _rel(g, gref)
    l2 = float(np.linalg.norm(g - gref) / max(1e-12, np.linalg.norm(gref)))
    flag = ' <<<<' if r > 5e-3 else ''
    print('%-50s rel %.2e l2 %.2e |ref| %.2e%s' % (p.name, r, l2, np.abs(gref).max(), flag))
