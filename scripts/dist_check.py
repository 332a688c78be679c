"""single-rank check of the data-parallel plumbing: with DL3P_FORCE_DIST=1 (RCCL collectives captured into the graphs,
deferred weight gradients) the loss trajectory must equal the plain single-GPU run bit for bit"""
import importlib, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    if os.environ.get('DL3P_FORCE_DIST'):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
    model = pkg.get_deeplabv3p_model('mobilenetv2', 21, (129, 129), 16, training=True)
    model.compile(optimizer=pkg.SGD(0.05, momentum=0.9), loss=pkg.SparseCategoricalCrossEntropy(ignore_index=255))
    gen = torch.Generator(device='cuda'); gen.manual_seed(7)
    x = torch.rand((4, 129, 129, 3), device='cuda', generator=gen) * 2 - 1
    y = torch.randint(0, 21, (4, 129 * 129, 1), device='cuda', generator=gen).float()
    ex = model._executor(4, True); ex.set_inputs(x, y); ex.lr.fill_(0.05)
    ex.train_step(); ex.capture()
    losses = []
    for _ in range(5):
        ex.train_step(); losses.append(float(ex.loss.item()))
    print('LOSSES ' + json.dumps(losses))
    if os.environ.get('DL3P_FORCE_DIST'):
        dist.destroy_process_group()
else:
    out = {}
    for tag, env in (('single', {}), ('forced_dist', {'DL3P_FORCE_DIST': '1'}),
                     ('forced_dist_segmented', {'DL3P_FORCE_DIST': '1', 'DL3P_COLLECTIVES_IN_GRAPH': '0'})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, 'child'], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('LOSSES')]
        if not line:
            print(tag, 'FAILED', r.stderr[-2000:]); sys.exit(1)
        out[tag] = json.loads(line[0][7:])
        print(tag, out[tag])
    assert out['single'] == out['forced_dist'] == out['forced_dist_segmented'], 'trajectories differ'
    print('OK identical')
