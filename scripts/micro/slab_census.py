"""Weight-gradient slab volume of one headline step, by layer: rows x n floats written by the weight-gradient kernel and read back by
dl3p_reduce_rows_batched (DESIGN 4a "one reduction for all weight gradients").  GPU box:  python scripts/micro/slab_census.py [model]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
mt = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
N = int(os.environ.get('DL3P_ST_N', 16))
model = pkg.get_deeplabv3p_model(mt, 21, (513, 513), 16, freeze_level=0, training=True)
model.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
ex = model._executor(N, True)
rec = ex._wgrad_jobs.cpu().numpy().view(np.dtype([('src', '<u8'), ('dst', '<u8'), ('rows', '<i4'), ('n', '<i4')]))
G0 = ex.store.G.data_ptr()
names = {}
for lay in ex.store.layers if hasattr(ex.store, 'layers') else []:
    pass
tot = 0
out = []
for r in rec:
    b = int(r['rows']) * int(r['n']) * 4
    tot += b
    out.append((b, int(r['rows']), int(r['n']), (int(r['dst']) - G0) // 4))
out.sort(reverse=True)
print('%d jobs, %.1f MB of slabs' % (len(out), tot / 1e6))
L = ex.L
for v in (0, 1):
    sel = [o for o in out if L.reduce_rows_variant(o[1], o[2]) == v]
    print('  reducer variant %d: %d jobs, %.1f MB, rows %s' % (v, len(sel), sum(o[0] for o in sel) / 1e6, sorted({o[1] for o in sel})))
for b, rows, n, off in out[:40]:
    print('%8.2f MB  rows %5d  n %8d  grad offset %d' % (b / 1e6, rows, n, off))
