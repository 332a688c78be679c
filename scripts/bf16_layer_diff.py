"""layer-by-layer comparison of the bf16 HIP forward with the rounding oracle (debugging aid):
python scripts/bf16_layer_diff.py [model_type] [H] [W]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle.np_net import OracleModel
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
mt = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
H = int(sys.argv[2]) if len(sys.argv) > 2 else 65
W = int(sys.argv[3]) if len(sys.argv) > 3 else 65
N, C = 2, 19
mp = pkg.mixed_precision
mp.set_policy(mp.Policy('mixed_bfloat16'))
m = pkg.get_deeplabv3p_model(mt, C, (H, W), 16, training=True)
mp.set_policy(mp.Policy('float32'))
m.compile(optimizer=pkg.SGD(0.01), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
o = OracleModel(mt, C, (H, W), 16, dtype=np.float64, seed=0)
o.net.bf16 = True
o.net.record = {}
m.set_weights_by_name(dict(o.net.params))
m.use_graphs = False
rng = np.random.default_rng(3)
x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
m.train_on_batch(x, y)
ex = m._executor(N, True)
drop = [op for op in m.graph.ops if op.kind == 'materialize' and op.rate > 0][0]
mask = ex.dropout_mask(drop).cpu().numpy()
o.loss_and_grads(x, y, {'aspp_dropout': mask})
for op in m.graph.ops:
    if op.kind in ('conv_pw', 'conv_dense', 'conv_dw') and op.name in o.net.record:
        ref = o.net.record[op.name]
        got = ex.view(op.out).float().cpu().numpy()[..., :ref.shape[-1]]
        err = np.abs(got - ref)
        ulp = 2.0 ** -8 * np.maximum(np.abs(ref), 1e-30)
        print('%-44s max|ref| %8.3f  max err %9.5f  rel-to-max %8.5f  frac>1ulp %.4f  frac>2ulp %.4f' % (
            op.name, np.abs(ref).max(), err.max(), err.max() / max(1e-9, np.abs(ref).max()), (err > 1.01 * ulp).mean(), (err > 2.01 * ulp).mean()))
