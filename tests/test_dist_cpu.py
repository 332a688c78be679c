"""Multi-process (world_size 2, gloo, CPU) test of the data-parallel protocol the HIP executor runs over
RCCL: per-rank local loss (mean over the local batch), SyncBatchNorm statistics all-reduced in forward
and backward, gradients all-reduced (sum) and scaled by 1/world -> identical to ONE process training on
the concatenated batch.  The compute here is the CPU oracle (allowed in tests only); the collectives,
the DistContext wrapper and the averaging convention are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(N, H, W, C):
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (N, H, W, 3))
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float64)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    return x, y


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle.np_net import OracleModel
        model_mod = load_pkg('model')
        ctx = model_mod.DistContext(sync_bn=True)
        assert ctx.world_size == world and ctx.rank == rank
        H = W = 33
        C, Nl = 5, 2
        x, y = _data(Nl * world, H, W, C)
        o = OracleModel('mobilenetv2_lite', C, (H, W), 16, dtype=np.float64, seed=0)

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            ctx.all_reduce(t)
            return t.numpy()
        o.net.sync = (allreduce, world)
        xs, ys = x[rank * Nl:(rank + 1) * Nl], y[rank * Nl:(rank + 1) * Nl]
        total, ce, _ = o.loss_and_grads(xs, ys)
        # gradient all-reduce over ONE flat buffer, then the 1/world scale the SGD kernel applies
        names = o.trainable_param_names()
        flat = torch.from_numpy(np.concatenate([o.net.grads[n].reshape(-1) for n in names]))
        ctx.all_reduce(flat)
        flat = flat.numpy() / world
        off = 0
        for n in names:
            sz = o.net.grads[n].size
            o.net.grads[n] = flat[off:off + sz].reshape(o.net.grads[n].shape)
            off += sz
        o.sgd_step(0.01, 0.9)
        losses = torch.tensor([ce], dtype=torch.float64)
        ctx.all_reduce(losses)
        if rank == 0:
            q.put(({n: o.net.params[n] for n in o.net.order}, float(losses.item()) / world))
    finally:
        dist.destroy_process_group()


def test_two_rank_syncbn_dp_equals_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    params, loss = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle.np_net import OracleModel
    H = W = 33
    C = 5
    x, y = _data(4, H, W, C)
    ref = OracleModel('mobilenetv2_lite', C, (H, W), 16, dtype=np.float64, seed=0)
    total, ce = ref.train_step(x, y)
    assert abs(loss - ce) < 1e-12 * max(1, abs(ce))
    for n in ref.net.order:
        np.testing.assert_allclose(params[n], ref.net.params[n], rtol=1e-9, atol=1e-11, err_msg=n)


def test_bench_reads_torchrun_env(monkeypatch):
    """bench.py takes RANK/LOCAL_RANK/WORLD_SIZE from the environment (driver contract)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    monkeypatch.setattr('sys.argv', ['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1'])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup, a.batch, a.size) == (2, 3, 1, 16, 513)
