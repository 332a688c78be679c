"""The data-parallel plumbing of the HIP executor on the one GPU a test box has (VERDICT r01 missing 2): with
DL3P_FORCE_DIST=1 a single rank runs the whole multi-rank path -- RCCL process group, SyncBatchNorm statistics
all-reduced on their own communicator, gradient buckets all-reduced on the side stream at the edges
`executor.bucket_edges` computes, deferred weight gradients, collectives captured INTO the hipGraphs (and the segmented
fallback) -- and must reproduce the plain single-GPU loss trajectory bit for bit (a one-rank all-reduce is the
identity, so any difference is an ordering / race / coverage bug of the plumbing).  More than one rank needs more than
one GPU: that run is the driver's (bench.py --gpus N)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_single_rank_dist_is_bit_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'dist_check.py')], capture_output=True, text=True,
                       cwd=ROOT, timeout=900)
    assert r.returncode == 0 and 'OK identical' in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_bench_runs_under_the_launcher_with_one_rank():
    """the driver's N > 1 command line, with N = 1: torch.distributed.run + RANK/LOCAL_RANK/WORLD_SIZE from the env"""
    import json
    env = dict(os.environ, DL3P_FORCE_DIST='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--size', '129', '--batch', '4', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads(lines[-1])
    # 65 BatchNorms x (forward + backward) + 4 gradient buckets = 134 un-coalesced; the ASPP branches share all-reduces
    assert out['n_gpus'] == 1 and out['value'] > 0 and 0 < out['config']['collectives_per_step'] <= 124
