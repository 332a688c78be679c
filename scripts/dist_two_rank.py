"""Two ranks against one (reference: train.py:143-158 MirroredStrategy = synchronous data parallelism).

  worker (under `python -m torch.distributed.run --nproc-per-node W`, or plain for W = 1):
      python scripts/dist_two_rank.py worker <model> <out.npz>
      trains <model> at 65 x 65 for 2 steps on ONE fixed batch of 4 images: rank r of W takes images [r 4/W, (r + 1) 4/W);
      dropout off (its mask is seeded per rank, so a 1-rank and a 2-rank run would draw different masks for the same image);
      rank 0 writes the step losses (mean over ranks = mean over the batch) and every weight to <out.npz>.
  With SyncBatchNorm over the global batch, gradients summed and scaled by 1 / W, the W = 2 trajectory equals the W = 1 one to
  rounding -- tests/test_dist_gpu.py::test_two_ranks_equal_one_rank_on_two_devices compares them whenever two devices are visible."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(model_type, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
    H = W = 65
    C, N = 21, 4
    model = pkg.get_deeplabv3p_model(model_type, C, (H, W), 16, training=True)
    for op in model.graph.ops:
        if op.kind == 'materialize' and getattr(op, 'rate', 0) > 0:
            op.rate = 0.0
    model.compile(optimizer=pkg.SGD(0.02, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255),
                  distributed=world > 1)
    model.use_graphs = os.environ.get('DL3P_TWO_RANK_GRAPHS', '1') != '0'
    rng = np.random.default_rng(17)
    x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
    y = rng.integers(0, C, (N, H * W, 1)).astype(np.float32)
    y[rng.uniform(size=y.shape) < 0.05] = 255
    n = N // world
    xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
    losses = []
    for _ in range(3):          # step 1 eager, step 2 captures, step 3 replays
        l = torch.tensor([model.train_on_batch(xs, ys)], dtype=torch.float64, device='cuda')
        if world > 1:
            dist.all_reduce(l)
        losses.append(float(l.item()) / world)
    torch.cuda.synchronize()
    ex = model._executor(n, True)
    info = dict(world=world, collectives=sum(getattr(pl, 'n_collectives', 0) for pl in (ex.fwd, ex.bwd, ex.opt)), graphed=bool(ex.graphed))
    if rank == 0:
        w = model.get_weights_by_name()
        np.savez(out, losses=np.array(losses), info=np.array([info['world'], info['collectives'], int(info['graphed'])]),
                 **{'w:' + k: v for k, v in w.items()})
        print('DIST_TWO_RANK', info, losses, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    if len(sys.argv) >= 4 and sys.argv[1] == 'worker':
        worker(sys.argv[2], sys.argv[3])
    else:
        print(__doc__)
        sys.exit(2)
