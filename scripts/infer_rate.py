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

# evaluation step: forward without the probability tensor + device-side argmax / confusion matrix
C = 21
y = torch.randint(0, C, (N, H * W, 1), device='cuda').float()
ex.set_inputs(x, y)
cm = torch.zeros(C * C, dtype=torch.int64, device='cuda')
for _ in range(3):
    ex.eval_step(cm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    ex.eval_step(cm)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print('%s eval_step batch %d: %.2f ms (forward + argmax + confusion matrix on the device), %.0f images/s' % (name, N, dt * 1e3, N / dt))
t0 = time.perf_counter()
for _ in range(5):
    p = model.predict(x.cpu().numpy())
    p.argmax(-1)
dt = (time.perf_counter() - t0) / 5
print('%s predict() + host argmax batch %d: %.1f ms, %.0f images/s' % (name, N, dt * 1e3, N / dt))

