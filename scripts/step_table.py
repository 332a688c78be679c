#!/usr/bin/env python3
"""Per-launch table of one training step: every C-ABI call of the traced plans, its kernels' GPU time
(from a rocprofv3 kernel trace) and the op it belongs to.

  run    (on the GPU box, under rocprofv3 --kernel-trace):
         rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -- python3 scripts/step_table.py run
         the step is replayed eagerly with a marker kernel (increment_kernel on a scratch counter)
         after every plan item; gpurun_out/st_labels.json holds the labels
  parse  python3 scripts/step_table.py parse gpurun_out/st > gpurun_out/step_table.txt
"""
import csv
import glob
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'


def run(model_type='mobilenetv2', N=None, size=513, C=21):
    # DL3P_ST_{N,H,W,C,OS,DTYPE}: other shapes / output stride / the bf16 policy (configs[3]: xception N=2 H=769 C=19 OS=8;
    # configs[4]: mobilenetv3large H=1024 W=2048 N=1 C=19 DTYPE=bf16)
    env = os.environ.get
    N = int(env('DL3P_ST_N', 0)) or N or (4 if model_type == 'xception' else 16)
    H, W = int(env('DL3P_ST_H', size)), int(env('DL3P_ST_W', env('DL3P_ST_H', size)))
    C = int(env('DL3P_ST_C', C))
    bf16 = env('DL3P_ST_DTYPE', 'f32') == 'bf16'
    import torch
    pkg = importlib.import_module(PKG)
    lib = importlib.import_module(PKG + '._lib').lib()
    if bf16:
        pkg.mixed_precision.set_policy(pkg.mixed_precision.Policy('mixed_bfloat16'))
    model = pkg.get_deeplabv3p_model(model_type, C, (H, W), int(env('DL3P_ST_OS', 16)), freeze_level=0, training=True)
    model.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234)
    x = torch.rand((N, H, W, 3), device='cuda', generator=gen) * 2 - 1
    y = torch.randint(0, C, (N, H * W, 1), device='cuda', generator=gen).float()
    ex = model._executor(N, True)
    ex.set_inputs(x, y)
    ex.lr.fill_(0.01)
    for _ in range(2):
        ex.train_step()
    torch.cuda.synchronize()
    scratch = torch.zeros(4, dtype=torch.int64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    labels = []
    for rep in range(3):                # the parser takes the last repetition
        labels = []
        lib.increment_counter(scratch.data_ptr(), st)
        lib.increment_counter(scratch.data_ptr(), st)      # double marker = start of a repetition
        for pname, plan in (('fwd', ex.fwd), ('bwd', ex.bwd), ('opt', ex.opt)):
            for (fn, args), (ep, ctx) in zip(plan.items, plan.labels):
                if fn is None:
                    args()
                else:
                    fn(*args, st)
                lib.increment_counter(scratch.data_ptr(), st)
                labels.append([pname, ep, ctx])
        torch.cuda.synchronize()
    exe = importlib.import_module(PKG + '.executor')
    shapes = {}
    for op in model.graph.ops:
        lab = exe._op_label(op)
        if op.kind in ('conv_pw', 'conv_dense'):
            shapes[lab] = dict(M=N * op.Ho * op.Wo, K=op.cin if op.kind == 'conv_pw' else op.kp, N=op.cout)
        elif op.kind == 'conv_dw':
            shapes[lab] = dict(Min=N * op.x.tensor.H * op.x.tensor.W, M=N * op.Ho * op.Wo, C=op.c, k=op.k, rate=op.rate,
                               stride=op.stride)
        elif op.kind == 'bn':
            shapes[lab] = dict(M=N * op.z.H * op.z.W, C=op.bn.C)
        elif getattr(op, 'out', None) is not None:
            shapes[lab] = dict(M=N * op.out.H * op.out.W, C=op.out.C)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    shapes['__esize__'] = 2 if bf16 else 4
    json.dump(shapes, open(os.path.join(ROOT, 'gpurun_out', 'st_shapes.json'), 'w'))
    json.dump(labels, open(os.path.join(ROOT, 'gpurun_out', 'st_labels.json'), 'w'))


def parse(d):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    labels = json.load(open(os.path.join(ROOT, 'gpurun_out', 'st_labels.json')))
    is_mark = ['increment_kernel' in r['Kernel_Name'] for r in rows]
    # a repetition starts with a run of >= 3 consecutive increment kernels: [previous rep's last marker,] the
    # double marker, the step's own increment_counter (item 0) and its marker
    runs, j = [], 0
    while j < len(rows):
        if is_mark[j]:
            e = j
            while e + 1 < len(rows) and is_mark[e + 1]:
                e += 1
            if e - j + 1 >= 3:
                runs.append(e)
            j = e + 1
        else:
            j += 1
    i = runs[-1] - 1
    us = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out, tot = [], 0.0
    for pname, ep, ctx in labels:
        ks = []
        if ep == 'dl3p_increment_counter' or ep == 'increment_counter':
            ks.append(rows[i]); i += 1                      # the item itself is an increment kernel
        else:
            while i < len(rows) and not is_mark[i]:
                ks.append(rows[i]); i += 1
        i += 1                                              # the marker
        t = sum(us(r) for r in ks)
        tot += t
        out.append((pname, ep, ctx, t, ['%s %.1f' % (r['Kernel_Name'].split('(')[0][:40], us(r)) for r in ks]))
    shapes = json.load(open(os.path.join(ROOT, 'gpurun_out', 'st_shapes.json')))
    ES = shapes.get('__esize__', 4)     # bytes per stored activation element (2 on the bf16 path)
    HBM, MFMA = 6.3e6, (157e6 if ES == 4 else 2500e6)   # achievable bytes/us (guide: ~6.3 TB/s), dense MFMA flop/us of the dtype

    pw, dw, bn = ('M', 'K', 'N'), ('Min', 'M', 'C'), ('M', 'C')
    NEED = dict(pwconv_fwd=pw, pwconv_fwd_wt=pw, pwconv_bwd_data=pw, pwconv_bwd_data_bn=pw, pwconv_bwd_weight=pw,
                dwconv2d_fwd=dw, dwconv2d_bwd_data=dw, dwconv2d_bwd_data_bn=dw, dwconv2d_bwd_weight=dw,
                bn_bwd_reduce=bn, bn_bwd_apply=bn)

    def floor_us(ep, ctx):
        sh = shapes.get(ctx)
        if not sh:
            return None
        ep = ep.replace('dl3p_', '').replace('_bf16', '')
        if not all(k in sh for k in NEED.get(ep, ())):
            return None
        if ep in ('pwconv_fwd', 'pwconv_fwd_wt', 'pwconv_bwd_data', 'pwconv_bwd_data_bn', 'pwconv_bwd_weight'):
            by = ES * sh['M'] * (sh['K'] + sh['N'])
            return max(by / HBM, 2.0 * sh['M'] * sh['K'] * sh['N'] / MFMA)
        if ep in ('dwconv2d_fwd', 'dwconv2d_bwd_data', 'dwconv2d_bwd_data_bn', 'dwconv2d_bwd_weight'):
            return ES * (sh['Min'] + sh['M']) * sh['C'] / HBM
        if ep == 'bn_bwd_reduce':
            return 2.0 * ES * sh['M'] * sh['C'] / HBM
        if ep == 'bn_bwd_apply':
            return 3.0 * ES * sh['M'] * sh['C'] / HBM
        return None
    print('# %d items, %.3f ms of kernel time' % (len(out), tot / 1e3))
    slack = {}
    for pname, ep, ctx, t, ks in out:
        fl = floor_us(ep, ctx)
        if fl:
            a = slack.setdefault((pname, ep), [0.0, 0.0])
            a[0] += t; a[1] += fl
        print('%-4s %-20s %-40s %8.1f us %s  %s' % (pname, ep.replace('dl3p_', ''), ctx, t,
              ('floor %6.1f (%3.0f%%)' % (fl, 100 * fl / t)) if fl else ' ' * 19, ' + '.join(ks)))
    print('\n# time vs floor (max(bytes / 6.3 TB/s, flops / 157 TF)) by entry point')
    for (pname, ep), (t, fl) in sorted(slack.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
        print('%-4s %-28s %9.1f us  floor %9.1f us  slack %9.1f us' % (pname, ep.replace('dl3p_', ''), t, fl, t - fl))
    # by entry point
    agg = {}
    for pname, ep, ctx, t, ks in out:
        a = agg.setdefault((pname, ep), [0, 0.0])
        a[0] += 1; a[1] += t
    print('\n# by entry point')
    for (pname, ep), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-4s %-28s n=%4d  %9.1f us  %5.1f%%' % (pname, ep.replace('dl3p_', ''), n, t, 100 * t / tot))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(*sys.argv[2:3])
    else:
        parse(sys.argv[2])
