import torch, time
def t(f, R=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e-3
for mb in (64, 406, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    tf = t(lambda: y.fill_(1.0)); tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum()); ta = t(lambda: torch.add(x, 1.0, out=y))
    print('%5d MB: fill %.2f TB/s  copy %.2f TB/s (r+w)  sum %.2f TB/s  add-out %.2f TB/s (r+w)' % (mb, 4*n/tf/1e12, 8*n/tc/1e12, 4*n/ts/1e12, 8*n/ta/1e12))
