import csv, glob, sys, importlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'increment_kernel' in r['Kernel_Name']]
last = rows[idx[-1]:]
us = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
m = importlib.import_module('tf-keras-deeplabv3p-model-set_amd.model')
model = m.get_deeplabv3p_model('mobilenetv2', 21, (513, 513), 16)
ops = [o for o in model.graph.ops if o.kind == 'conv_pw']
import re
def bkn(r):
    mm = re.search(r'pw_gemm_kernel<(\d+), (true|false), (true|false)', r['Kernel_Name'])
    return mm.group(2) if mm else None
fw = [r for r in last if bkn(r) == 'true']
dg = [r for r in last if bkn(r) == 'false']
wg = [r for r in last if 'pw_wgrad_kernel' in r['Kernel_Name']]
print(len(ops), len(fw), len(dg), len(wg))
N = 16
tot = [0, 0, 0]
dgi = iter(dg)
table = []
for op, rf, rw in zip(ops, fw, reversed(wg)):
    pass
dgl = list(reversed(dg))
# dgrad exists for every op whose input needs grad (all but none here) -> align from the end
k = 0
for i, op in enumerate(ops):
    M = N * op.Ho * op.Wo
    gf = 2.0 * M * op.cin * op.cout / 1e9
    t_f = us(fw[i]); t_w = us(wg[len(ops) - 1 - i])
    t_d = us(dgl[i]) if len(dgl) == len(ops) else float('nan')
    tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
    print('%-26s M=%7d K=%4d N=%4d fwd %6.1f us %5.1f TF | dgrad %6.1f us %5.1f TF | wgrad %6.1f us %5.1f TF' % (
        op.name, M, op.cin, op.cout, t_f, gf / t_f * 1e3, t_d, gf / t_d * 1e3, t_w, gf / t_w * 1e3))
print('totals us', tot)
