"""Inference (predict-shaped forward: frozen BN folded into the consumers' prologues, softmax probabilities out)
images/s with the batch resident, eager launches vs hipGraph replay.  usage: python scripts/infer_rate.py [model] [batch]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module('tf-keras-deeplabv3p-model-set_amd')
name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = W = 513
model = pkg.get_deeplabv3p_model(name, 21, (H, W), 16, training=False)
ex = model._executor(N, False)
x = torch.rand((N, H, W, 3), device='cuda') * 2 - 1
ex.set_inputs(x)


def rate(tag, reps=30):
    for _ in range(3):
        ex.forward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.forward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%s %s batch %d: %.2f ms per forward, %.0f images/s (%d launches)' % (name, tag, N, dt * 1e3, N / dt, ex.fwd.n_launches))


rate('eager ')
ex.capture()
rate('graph ')
