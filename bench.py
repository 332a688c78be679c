#!/usr/bin/env python3
"""Headline benchmark: images/sec of the MobileNetV2-DeepLabV3+ (ASPP 6/12/18 + decoder) training step,
513x513x3 synthetic batches, 21 classes, OS=16, fp32, per-GPU batch 16 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
   one rank per GPU over RCCL; weak scaling: per-GPU batch fixed)

A step = forward + softmax-CE(ignore 255) + L2 + backward + SGD(momentum) over one resident batch,
replayed as hipGraph segments.  Rank 0 prints ONE JSON line.  Besides the throughput it carries
  roofline     : the rate-18 atrous depthwise kernel (ASPP aspp3_depthwise) timed live with HIP events
                 inside the timed steps, priced against its ALGORITHMIC bytes
  cpu_baseline : the NumPy oracle (a CPU port of the same train step, BASELINE.json configs[0]:
                 mobilenetv2_lite, batch 2) timed on this box's host cores, rank 0 / N == 1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'tf-keras-deeplabv3p-model-set_amd'
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: dense fp32 matrix peak (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16 (the split-bf16 GEMM issues 6 bf16 products per fp32 product)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)      # warm-up 20, >= 100 timed steps
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch (with --scaling strong: the GLOBAL batch)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="weak: per-GPU batch fixed (default, what the driver measures); strong: the reference's semantics, "
                         "--batch is the global batch split across the ranks (train.py:143-158)")
    ap.add_argument('--size', type=int, default=513, help='input height (and width unless --width is given)')
    ap.add_argument('--width', type=int, default=0, help='input width for non-square inputs (1024x2048 Cityscapes)')
    ap.add_argument('--model', default='mobilenetv2')
    ap.add_argument('--classes', type=int, default=21)
    ap.add_argument('--os', type=int, default=16, help='output stride')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="bf16: the mixed-precision policy of train.py:37-46 (bf16 storage, fp32 accumulate; BASELINE configs[4])")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-sync-bn', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-streaming', action='store_true', help='skip the N=256 DRAM-streaming run of the roofline kernel')
    ap.add_argument('--cpu-steps', type=int, default=10)
    ap.add_argument('--split-gemm', type=int, default=None, choices=[0, 1],
                    help='1: compute-bound 1x1 convs as fp32-accurate split-bf16 GEMMs on the bf16 matrix pipe (csrc/pw_split.hip); '
                         '0: fp32-input MFMA everywhere; default: DL3P_SPLIT_GEMM or 1')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short runs of BASELINE.json configs[2..4] behind the headline region')
    return ap.parse_args()


def head_name(model_type, OS):
    """what sits on the backbone: ASPP_block + Decoder_block (layers.py:114-219) or ASPP_Lite_block (:166-196) for the *_lite types"""
    if model_type.endswith('_lite'):
        return 'ASPP-Lite (image pooling + 1x1), no decoder'
    return 'ASPP(%s) + decoder' % {8: '12/24/36', 16: '6/12/18', 32: '3/6/9'}[OS]


def cpu_baseline(args):
    """the CPU stand-in for the reference's tf.keras train.py (which cannot be installed here):
    the same graph (configs[0]: mobilenetv2_lite, 513x513, batch 2, fwd + CE(ignore 255) + L2 + bwd + SGD momentum,
    fp32) restated with torch-CPU ops (oneDNN) on all host cores -- oracle/torch_net.py, the checker's code, never the
    product path -- 3 warm-up + 10 timed steps; one core and the NumPy oracle are reported beside it"""
    import numpy as np
    import torch
    from oracle.torch_net import TorchModel
    H = W = args.size
    B, C = 2, args.classes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    rng = np.random.default_rng(1234)
    x = rng.uniform(-1, 1, (B, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (B, H * W, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    mask = (rng.uniform(size=(B, (H + 15) // 16, (W + 15) // 16, 256)) >= 0.5).astype(np.float32)

    def rate(threads, warm, steps, budget_s):
        torch.set_num_threads(threads)
        m = TorchModel('mobilenetv2_lite', C, (H, W), 16, dtype=np.float32, seed=0)
        for _ in range(warm):
            m.train_step(x, y, {'aspp_dropout': mask})
        t0, n = time.time(), 0
        while n < steps and (n == 0 or time.time() - t0 < budget_s):
            m.train_step(x, y, {'aspp_dropout': mask})
            n += 1
        return B * n / (time.time() - t0), n
    # the thread count is MEASURED, not guessed: a short probe at 8 / 16 / 32 / 64 threads, then the timed
    # sample at the best of them.  (oneDNN does not scale on a batch-2 graph: on the 256-core GPU host 32 threads gave 6.8
    # images/s, 64 threads 2.8, 128 threads 1.2 and all 256 cores 0.011 -- three minutes per step -- so the probe stops at 64.)
    probe = {}
    for th in sorted({t for t in (8, 16, 32, 64) if t <= cores}):
        probe[th] = rate(th, 1, 4, 8.0)[0]      # (>= 4 timed steps per candidate: with two the probe and the sample disagreed by 40 %)
    threads = max(probe, key=probe.get)
    all_rate, all_n = rate(threads, 3, args.cpu_steps, 30.0)
    one_rate, one_n = rate(1, 1, 3, 15.0)
    out = {'value': round(all_rate, 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
           'sample': '%d train steps (3 warm-up) of mobilenetv2_lite %dx%d batch %d fp32: torch-CPU (oneDNN) restatement of the '
                     'same graph, NOT tf.keras (not installable here); host has %d cores' % (all_n, H, W, B, cores),
           'thread_probe': {str(k): round(v, 3) for k, v in probe.items()}, 'thread_probe_steps': 4,
           'one_core': {'value': round(one_rate, 3), 'steps': one_n}}
    try:
        from oracle.np_net import OracleModel
        o = OracleModel('mobilenetv2_lite', C, (H, W), 16, dtype=np.float32, seed=0)
        o.train_step(x, y, {'aspp_dropout': mask})
        t0 = time.time()
        o.train_step(x, y, {'aspp_dropout': mask})
        out['numpy_oracle'] = {'value': round(B / (time.time() - t0), 3), 'steps': 1, 'cores': cores}
    except Exception as e:      # noqa: BLE001 - the second figure is optional
        out['numpy_oracle'] = {'error': str(e)[:80]}
    return out


def streaming_variant(pkg, op):
    """the roofline kernel on a tensor that cannot be cache-resident (N = 256: 357 MB in, 357 MB out
    against 256 MB of Infinity Cache), outside the step, timed per launch with the library's HIP event pair"""
    import ctypes
    import torch
    ops = importlib.import_module(PKG + '.ops')
    L = importlib.import_module(PKG + '._lib').lib()
    t = op.out
    Nb, H, W, C = 256, t.H, t.W, op.c
    x = torch.randn((Nb, H, W, C), device='cuda')
    y = torch.empty_like(x)
    w = torch.randn((op.k, op.k, C), device='cuda')
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    part = ops.new_partials(C, 'cuda')
    ts = []
    for i in range(12):
        L.probe_arm(4000 + i)
        ops.dwconv2d_fwd(x, w, 1, op.rate, 'same', sc, sh, ops.ACT_RELU, out=y, partials=part)
    torch.cuda.synchronize()
    for i in range(2, 12):
        ms = ctypes.c_float(0)
        L.probe_read(4000 + i, ctypes.addressof(ms))
        ts.append(ms.value)
    us = 1e3 * sum(ts) / len(ts)
    by = 2.0 * Nb * H * W * C * 4 + op.k * op.k * C * 4
    return {'streaming_frac': round(by / us / 1e3 / HBM_PEAK_GBS, 4), 'streaming_avg_us': round(us, 2),
            'streaming_shape': 'N=%d %dx%dx%d (%.0f MB each way)' % (Nb, H, W, C, by / 2e6)}


def kernel_source_hash():
    """what the roofline kernel is built from: the depthwise kernels and their measured plan table (git blob hashes, as
    `git hash-object` prints them).  scripts/collect_profiles.sh stamps the PMC record with it; a record collected on other
    sources is refused below instead of surviving a kernel change silently"""
    import hashlib
    out = []
    for f in ('csrc/dwconv.hip', 'csrc/dw_tuned.h'):
        data = open(os.path.join(ROOT, PKG, f), 'rb').read()
        out.append(hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()[:12])
    return '+'.join(out)


def measured_traffic(kernel_name):
    """HBM bytes per launch of the roofline kernel from the newest committed PMC pass (profiles/*_roofline_traffic.json,
    written by scripts/collect_profiles.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied).
    -> (bytes | None, source file | reason)"""
    import glob
    want = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_roofline_traffic.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get('kernel') and d['kernel'] in (kernel_name or ''):
            if d.get('source_hash') != want:
                return None, '%s was collected on other kernel sources (%s, now %s): re-run scripts/collect_profiles.sh' % (
                    os.path.basename(f), d.get('source_hash', 'unstamped'), want)
            return int(d['traffic_bytes']), os.path.basename(f)
    return None, None


# BASELINE.json configs[2..4] at their per-GPU shapes (C3 Xception 513x513 batch 32 over 8 GPUs -> 4 per
# GPU; C4 Xception 769x769 OS 8, 19 classes, batch 16 over 8 -> 2; C5 MobileNetV3-Large 1024x2048, 19 classes, bf16, batch 8
# over 8 -> 1): (tag, model, classes, H, W, output stride, per-GPU batch, dtype)
OTHER_CONFIGS = [('configs[2]', 'xception', 21, 513, 513, 16, 4, 'f32'),
                 ('configs[3]', 'xception', 19, 769, 769, 8, 2, 'f32'),
                 ('configs[4]', 'mobilenetv3large', 19, 1024, 2048, 16, 1, 'bf16')]


def other_config(pkg, tag, model_type, C, H, W, OS, N, dtype, steps=20, warmup=5):
    """a short timed loop of one of the other BASELINE configs on this GPU, AFTER the headline region (its numbers never
    enter `value`): the same step (fwd + loss + bwd + SGD, BN training, dropout), hipGraph replay, with the config's own top
    atrous kernel (the highest ASPP rate) between HIP events inside the steps"""
    import torch
    mp = pkg.mixed_precision
    mp.set_policy(mp.Policy('mixed_bfloat16' if dtype == 'bf16' else 'float32'))
    try:
        model = pkg.get_deeplabv3p_model(model_type, C, (H, W), OS, freeze_level=0, training=True)
    finally:
        mp.set_policy(mp.Policy('float32'))
    model.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    gen = torch.Generator(device='cuda')
    gen.manual_seed(4321)
    x = torch.rand((N, H, W, 3), device='cuda', generator=gen) * 2 - 1
    y = torch.randint(0, C, (N, H * W, 1), device='cuda', generator=gen).float()
    y[torch.rand(y.shape, device='cuda', generator=gen) < 0.05] = 255.0
    ex = model._executor(N, True)
    ex.set_inputs(x, y)
    ex.lr.fill_(0.01)
    probe = ex.install_probe('aspp3_depthwise')
    ex.train_step()
    ex.capture()
    for _ in range(warmup):
        ex.train_step()
    probe.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ex.train_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = float(ex.loss.item())
    op, t = probe.op, probe.op.out
    es = 2 if dtype == 'bf16' else 4
    algo = 2.0 * N * t.H * t.W * op.c * es + op.k * op.k * op.c * es
    us = probe.mean_ms() * 1e3
    out = {'config': tag,
           'workload': '%s + %s, OS=%d, %dx%d, %d classes, per-GPU batch %d, fwd+loss+bwd+SGD' % (
               model_type, head_name(model_type, OS), OS, H, W, C, N),
           'dtype': dtype, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(1000 * dt / steps, 3),
           'images_per_sec': round(N * steps / dt, 2), 'final_loss': round(loss, 5),
           'launches_per_step': ex.fwd.n_launches + ex.bwd.n_launches + ex.opt.n_launches,
           'roofline': {'bound': 'hbm', 'kernel': probe.kernel_name, 'avg_us': round(us, 3), 'algorithmic_bytes': int(algo),
                        'achieved': round(algo / (us * 1e-6) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(algo / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        'shape': 'N=%d %dx%dx%d k=%d rate=%d' % (N, t.H, t.W, op.c, op.k, op.rate)}}
    del ex, model, x, y
    torch.cuda.empty_cache()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run job (this
    process has not touched the GPU and never will), pass its output through and relay rank 0's JSON line as the last
    line.  The driver's own `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE and never gets here."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    js = [l for l in lines if l.startswith('{') and '"metric"' in l]
    for l in lines:
        if not js or l is not js[-1]:
            print(l)
    if js:
        print(js[-1], flush=True)
    return proc.returncode if proc.returncode else (0 if js else 1)


def main():
    args = parse()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (what RCCL needs here); before HIP starts
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import numpy as np
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (there is no CPU fallback for the HIP path)')
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get('DL3P_FORCE_DIST'):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if args.split_gemm is not None:
        os.environ['DL3P_SPLIT_GEMM'] = str(args.split_gemm)
    pkg = importlib.import_module(PKG)
    H, W = args.size, (args.width or args.size)
    N, C = args.batch, args.classes
    if args.scaling == 'strong':
        if args.batch % world:
            raise SystemExit('--scaling strong: the global batch %d does not divide over %d ranks' % (args.batch, world))
        N = args.batch // world

    if args.dtype == 'bf16':
        pkg.mixed_precision.set_policy(pkg.mixed_precision.Policy('mixed_bfloat16'))
    model = pkg.get_deeplabv3p_model(args.model, C, (H, W), args.os, freeze_level=0, training=True)
    model.compile(optimizer=pkg.SGD(0.01, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255),
                  sync_bn=not args.no_sync_bn)
    model.use_graphs = not args.no_graph

    # synthetic resident batch: U[-1,1) images, labels in [0,C) with 5 % set to 255
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1234 + rank)
    x = torch.rand((N, H, W, 3), device='cuda', generator=gen) * 2 - 1
    y = torch.randint(0, C, (N, H * W, 1), device='cuda', generator=gen).float()
    y[torch.rand(y.shape, device='cuda', generator=gen) < 0.05] = 255.0

    # world > 1: bound the part where a multi-rank job can hang (executor trace = every collective once, graph capture, first
    # replays); on expiry the guard prints the switches to try and exits non-zero (watchdog.py)
    guard = pkg.watchdog.FirstStepsGuard(rank, world, 'executor trace')
    guard.__enter__()
    try:
        ex = model._executor(N, True)
        ex.set_inputs(x, y)
        ex.lr.fill_(0.01)

        # roofline probe: the rate-18 depthwise launch stays outside the graph segments, between two events
        probe_name = 'aspp3_depthwise' if any(getattr(o, 'name', '') == 'aspp3_depthwise' for o in model.graph.ops) else None
        # (N == 1 only: with collectives captured into the graph the forward must stay one segment)
        dist_mode = world > 1 or os.environ.get('DL3P_FORCE_DIST', '0') not in ('', '0')
        probe = ex.install_probe(probe_name) if (probe_name and not dist_mode) else None
        # second probe: the largest pointwise GEMM of the step (decoder_conv0_pointwise) against the fp32 MFMA peak
        # (north_star: "MFMA utilisation for the pointwise GEMMs"); fp32 path only
        pw_name = 'decoder_conv0_pointwise' if any(getattr(o, 'name', '') == 'decoder_conv0_pointwise' for o in model.graph.ops) else None
        want_pw_probe = bool(pw_name and not dist_mode and args.dtype == 'f32')

        def barrier():
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        guard.stage('first eager step')
        ex.train_step()                      # first step eager (also the graph-capture warm-up)
        if model.use_graphs:
            guard.stage('graph capture')
            ex.capture()
        guard.stage('warm-up replays')
        for _ in range(args.warmup):
            ex.train_step()
        if probe:
            probe.reset()
        barrier()
    finally:
        guard.__exit__(None, None, None)      # always cancelled: an exception must not leave the timer armed
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.train_step()
    barrier()
    dt = time.perf_counter() - t0
    loss = float(ex.loss.item())
    pw_probe = None
    if want_pw_probe:
        # a few more steps AFTER the timed region with the GEMM launch between event pairs (the launch leaves the hipGraph
        # for that, which would cost the timed steps ~0.1 ms)
        pw_probe = ex.install_pw_probe(pw_name)
        if model.use_graphs:
            ex.capture()
        ex.train_step()
        pw_probe.reset()
        for _ in range(10):
            ex.train_step()
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        out = {
            'metric': ('images/sec (513x513, 21-class) MobileNetV2-DeepLabV3+ OS=16 training step'
                       if (args.model, H, W, C, args.os) == ('mobilenetv2', 513, 513, 21, 16) else
                       'images/sec (%dx%d, %d-class) %s-DeepLabV3+ OS=%d training step' % (H, W, C, args.model, args.os)),
            'value': round(N * world * args.steps / dt, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%s + %s, OS=%d, %dx%d, %d classes, per-GPU batch %d, '
                                   'fwd+loss+bwd+SGD(momentum 0.9, l2 2e-5), BN training mode, dropout 0.5'
                                   % (args.model, head_name(args.model, args.os), args.os, H, W, C, N),
                       'global_batch': N * world,
                       'parallelism': 'dp%d%s' % (world, '+syncbn' if (world > 1 and not args.no_sync_bn) else ''),
                       'hip_graph': bool(model.use_graphs), 'final_loss': round(loss, 5),
                       'gemm': ('fp32-accurate split-bf16 (3 x bf16 pieces, 6 products, fp32 accumulate) for the compute-bound 1x1 convs, '
                                'fp32-input MFMA elsewhere' if model._store.Sb is not None else 'fp32-input MFMA'),
                       'launches_per_step': ex.fwd.n_launches + ex.bwd.n_launches + ex.opt.n_launches,
                       'collectives_per_step': sum(getattr(pl, 'n_collectives', 0) for pl in (ex.fwd, ex.bwd, ex.opt)),
                       'rccl_ranks': (torch.distributed.get_world_size() if (world > 1 and torch.distributed.is_initialized()) else 1)},
        }
        standalone = None
        if not probe and probe_name and dist_mode and args.dtype == 'f32':
            # N > 1: the forward is one graph (collectives captured), so the roofline launch is re-issued on the buffers of
            # the last step and timed with the same HIP event pair
            class _P:
                pass
            probe = _P()
            probe.op = [o for o in model.graph.ops if getattr(o, 'name', '') == probe_name and o.kind == 'conv_dw'][0]
            probe.kernel_name = 'dw_fwd_lattice2' if (args.model, args.size, args.os) == ('mobilenetv2', 513, 16) else 'depthwise forward'
            # ... behind the launches that precede it in the step, from the GEMM that writes its input (so that it meets
            # the step's cache state, not one warmed by its own previous run)
            prod = ex.g.producer_of(probe.op.x.tensor)
            standalone = ex.time_tagged_launch(probe_name, context_from='pw:' + getattr(prod, 'name', ''))
        if probe:
            ms = standalone if standalone is not None else probe.mean_ms()
            op = probe.op
            t = op.out
            es = 2 if args.dtype == 'bf16' else 4
            algo = 2.0 * N * t.H * t.W * op.c * es + op.k * op.k * op.c * es     # read x once, write y once, weights
            ach = algo / (ms * 1e-3) / 1e9
            traffic, src = (measured_traffic(probe.kernel_name) if (args.size, N, args.model) == (513, 16, 'mobilenetv2')
                            else (None, None))
            out['roofline'] = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': src,
                               'kernel': probe.kernel_name, 'avg_us': round(ms * 1e3, 3),
                               'algorithmic_bytes': int(algo),
                               'shape': 'N=%d %dx%dx%d k=%d rate=%d' % (N, t.H, t.W, op.c, op.k, op.rate)}
            if standalone is not None:
                out['roofline']['measured'] = ('rank 0, after the timed region: 20 x [the launches of the step from the GEMM that writes its '
                                               'input up to it, re-issued on the last step\'s buffers], the last one timed')
        if pw_probe:
            ms = pw_probe.mean_ms()
            op = pw_probe.op
            M, K, Nc = N * op.Ho * op.Wo, op.cin, op.cout
            tf = 2.0 * M * K * Nc / (ms * 1e-3) / 1e12
            if model._store.Sb is not None and op in model._store.sb_fwd:
                # which split kernel the planner gives this launch (family 3: {3, nt, mi, wm, ...}; wm 4 = the pinned-schedule form)
                import ctypes as _ct
                q = (_ct.c_int * 6)()
                importlib.import_module(PKG + '._lib').lib().gemm_plan_query(6, M, K, Nc, q)
                sb_kernel, sb_mfma = (('pw_gemm_sb3_kernel', 'v_mfma_f32_32x32x16_bf16') if q[3] == 4 else
                                      ('pw_gemm_sb_kernel', 'v_mfma_f32_16x16x32_bf16'))
                # split-bf16 kernel: six bf16 x bf16 products per fp32 product on v_mfma_f32_16x16x32_bf16, priced against the
                # dense bf16 peak (the guide's 2.5 PFLOP/s; 1.5 PFLOP/s is what the pipe sustains on this instruction mix)
                out['roofline_mfma'] = {'bound': 'mfma', 'achieved': round(6 * tf, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                        'frac': round(6 * tf / MFMA_BF16_PEAK_TFLOPS, 4), 'kernel': '%s (%s forward)' % (sb_kernel, op.name),
                                        'avg_us': round(ms * 1e3, 2), 'flops': int(12.0 * M * K * Nc),
                                        'fp32_equivalent_tflops': round(tf, 2), 'vs_fp32_mfma_peak': round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                        'shape': 'M=%d K=%d N=%d, fp32 operands split into 3 bf16 pieces, 6 products (%s)' % (M, K, Nc, sb_mfma)}
            else:
                out['roofline_mfma'] = {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                        'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'kernel': 'pw_gemm_kernel (%s forward)' % op.name,
                                        'avg_us': round(ms * 1e3, 2), 'flops': int(2.0 * M * K * Nc),
                                        'shape': 'M=%d K=%d N=%d fp32 (v_mfma_f32_16x16x4_f32)' % (M, K, Nc)}
        if probe and standalone is None and args.dtype == 'f32' and not args.no_streaming:
            out['roofline'].update(streaming_variant(pkg, probe.op))
        headline = (args.model, H, W, C, args.os, args.dtype) == ('mobilenetv2', 513, 513, 21, 16, 'f32')
        if world == 1 and headline and not dist_mode and model._store.Sb is not None and not args.no_other_configs:
            # the same step with EVERY GEMM on the fp32-input MFMA kernels (DL3P_SPLIT_GEMM=0: the headline path of rounds 1-2;
            # DESIGN section 4c): a short loop behind the headline region, reported beside it
            try:
                os.environ['DL3P_SPLIT_GEMM'] = '0'
                out['fp32_mfma_only'] = other_config(pkg, 'configs[1], DL3P_SPLIT_GEMM=0', 'mobilenetv2', C, H, W, args.os, N, 'f32', steps=30, warmup=10)
                out['fp32_mfma_only']['gemm'] = 'fp32-input MFMA (v_mfma_f32_16x16x4_f32) for every 1x1 conv'
            except Exception as e:      # noqa: BLE001
                out['fp32_mfma_only'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
            finally:
                if args.split_gemm is None:
                    os.environ.pop('DL3P_SPLIT_GEMM', None)
                else:
                    os.environ['DL3P_SPLIT_GEMM'] = str(args.split_gemm)
        if world == 1 and headline and not dist_mode and not args.no_other_configs:
            # the other BASELINE configs, each a short loop behind the headline region
            del ex
            model._exec = {}
            torch.cuda.empty_cache()
            out['other_configs'] = []
            for cfg in OTHER_CONFIGS:
                try:
                    out['other_configs'].append(other_config(pkg, *cfg))
                except Exception as e:      # noqa: BLE001 -- never lose the headline line to a side run
                    out['other_configs'].append({'config': cfg[0], 'error': '%s: %s' % (type(e).__name__, str(e)[:300])})
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        # native libraries (RCCL's version banner) write through C stdio: flush their buffer first so that the JSON
        # line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
